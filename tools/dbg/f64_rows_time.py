"""Launch time of the strict-parity rollout (rollout_cost_rows_kernel / the thread form under ICEM_GK_ROLLOUT=thread) against the
horizon: separates the launch's fixed cost from its per-step cost.  usage (GPU box): python tools/dbg/f64_rows_time.py [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
for dtype in ("f64", "f32"):
    if dtype == "f32":
        os.environ["ICEM_DISABLE_FAST"] = "1"
    for h in (2, 10, 30, 60):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=6, num_traj=N, opt_iters=2, dtype=dtype, seed=1), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        act = torch.rand((N, h, 6), dtype=pl.dt, device=pl.device) * 2 - 1
        obs = 0.1 * np.random.RandomState(0).randn(17)
        for _ in range(3):
            pl.rollout_cost(obs, act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            pl.rollout_cost(obs, act)
        e1.record()
        torch.cuda.synchronize()
        print(f"{dtype} N={N} h={h:2d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us per launch (incl. the host's obs0 upload)")
