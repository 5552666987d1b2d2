"""us per MPC step (icem_plan_step, resident inputs) at the populations given on the command line; the c2 shapes by default,
others with --d / --o / --kind / --beta / --iters (e.g. `--d 17 --o 24 --kind 1 --beta 2 --iters 3` = the d = 17 latent
HumanoidStandup shape, `--o 18 --kind 1` = HalfCheetah with the x position)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env, humanoid_standup_env
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # ICEM_<NAME> variables (incl. ICEM_TILE_ARITH) are mapped per planner: the library reads no environment
ap = argparse.ArgumentParser()
ap.add_argument("N", type=int, nargs="*", default=[4096, 8192, 16384, 32768])
ap.add_argument("--d", type=int, default=6)
ap.add_argument("--o", type=int, default=17)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--beta", type=float, default=0.25)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--band", type=int, default=-1, help="0: diagonal A (SURVEY 7.3-11's cheap dynamics), -1: dense")
a = ap.parse_args()
env = halfcheetah_env(a.o) if a.d == 6 else humanoid_standup_env(a.o)
model = DeviceSyntheticModel.make(a.o, a.d, kind=a.kind, band=a.band)
for N in a.N:
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=a.d, num_traj=N, opt_iters=a.iters, dtype="f32", seed=1, noise_beta=a.beta),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(a.o), dtype=pl.dt))
    for _ in range(10):
        pl.plan_step_resident()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    print(f"N={N} d={a.d} o={a.o} kind={a.kind} band={a.band}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per MPC step", flush=True)
