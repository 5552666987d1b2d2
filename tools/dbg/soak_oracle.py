"""How often does the f32 device loop pick a different elite set than the float64 oracle on the same Philox draws?
(near-ties between trajectory costs: the f32 rollout differs from f64 by ~1e-6 relative).  Runs `n` seeds of a
3-iteration MPC step at N=1000 and reports the fraction whose executed action / refit mean leave the test tolerance."""
import sys

import numpy as np

sys.path.insert(0, ".")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env  # noqa: E402
from oracle import icem_oracle as O  # noqa: E402 (checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6, kind=1)
om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
N, h, d, iters = 1000, 30, 6, 3
bad = 0
for seed in range(n):
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype=dtype, seed=seed), env.action_space.low,
                     env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float32 if dtype == "f32" else np.float64)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64),
                       lambda ob, ac: O.rollout_costs(om, oc, ob, ac), lambda num: tuple(z.astype(np.float64) for z in noise(num)))
    orc.beginning_of_rollout()
    ok = True
    for s in range(2):
        ob = 0.1 * np.random.RandomState(1000 + seed * 7 + s).randn(17)
        if s:
            noise.begin_step()
        a = pl.plan_step(ob).cpu().numpy().astype(np.float64)
        w = orc.get_action(ob)
        ok &= np.allclose(a, w, rtol=2e-4, atol=2e-5) and np.allclose(pl.mean.cpu().numpy(), orc.mean, rtol=2e-4, atol=2e-5)
    bad += not ok
    if not ok:
        print("seed", seed, "differs", flush=True)
print(f"{dtype}: {bad} of {n} seeds leave the tolerance (elite set differs from the float64 oracle's)")
