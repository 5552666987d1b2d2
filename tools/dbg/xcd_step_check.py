"""The one-launch step of small populations (k_step_xcd.hip) against the launches-per-iteration path (option step_xcd = 0):
every buffer bit for bit over several MPC steps, then the time per MPC step of both."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
from icem_amd import _lib as L


def make(N, iters, kind=0, mode="sum", seed=5, arith=None, h=30, d=6, o=17):
    env = halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=seed, cost_mode=mode),
                     env.action_space.low[:d], env.action_space.high[:d])
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    if arith is not None:
        pl.set_tile_arith(arith)
    pl.reset()
    return pl


def state(pl):
    n_last = pl.population_sizes[-1]
    ea, ec = pl.current_elites()
    f = lambda t: t.detach().cpu().numpy().copy()
    return [f(pl.executed), f(pl.best_cost), f(pl.mean), f(pl.std), f(ea), f(ec), f(pl.costs[:n_last]), f(pl.actions[:n_last])]


names = ["executed", "best_cost", "mean", "std", "elites", "elite_costs", "costs", "actions"]
bad = 0
for (N, iters, kind, mode, arith) in [(4096, 5, 0, "sum", None), (4096, 5, 1, "best", None), (1000, 3, 0, "final", None), (4096, 2, 0, "sum", "f32"),
                                      (300, 4, 1, "sum", None), (2500, 1, 0, "sum", None)]:
    L.reset_options(); L.set_option("step_xcd", 0)
    ref = make(N, iters, kind, mode, arith=arith)
    L.set_option("step_xcd", 1)
    new = make(N, iters, kind, mode, arith=arith)
    for s in range(5):
        obs = 0.1 * np.random.RandomState(s).randn(17)
        L.set_option("step_xcd", 0); ref.plan_step(obs)
        L.set_option("step_xcd", 1); new.plan_step(obs)
        torch.cuda.synchronize()
        for nm, x, y in zip(names, state(new), state(ref)):
            if not np.array_equal(x, y, equal_nan=True):
                bad += 1
                print(f"MISMATCH N={N} iters={iters} kind={kind} {mode} step {s}: {nm}  max|diff|={np.nanmax(np.abs(x.astype(np.float64)-y.astype(np.float64))):.3e}", flush=True)
    print(f"N={N} iters={iters} kind={kind} {mode} arith={arith}: xcd launches, timed out = {new.step_status()}; ref = {ref.step_status()}", flush=True)
print("mismatches:", bad)
for on in (0, 1, 0, 1):
    L.set_option("step_xcd", on)
    pl = make(4096, 5)
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    for _ in range(30):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    print(f"step_xcd={on}: {1e6 * (time.perf_counter() - t0) / 300:.1f} us per MPC step   status {pl.step_status()}", flush=True)
