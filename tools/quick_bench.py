import time, numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17)
cfgs = [(4096, 5, -1), (65536, 1, -1)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for (N, iters, band) in cfgs:
    model = DeviceSyntheticModel.make(17, 6, band=band)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=iters, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    obs = 0.1*np.random.RandomState(0).randn(17)
    for _ in range(5): pl.plan_step(obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 20
    for _ in range(K): pl.plan_step(obs)
    torch.cuda.synchronize()
    dt = (time.perf_counter()-t0)/K
    trajsteps = sum(pl.population_sizes)*30
    print(f"N={N} iters={iters} band={band}: {dt*1e6:.1f} us/MPC-step, {trajsteps/dt/1e9:.2f} G traj-steps/s")
