import time, numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd import _lib as L
w = bench.WORKLOADS["c2"]; env = bench.make_env(w)
for B in (4, 8, 16):
    pls = []
    for i in range(B):
        model = DeviceSyntheticModel.make(w["o"], w["d"], seed_a=2*i, seed_b=2*i+1)
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=4096, opt_iters=5, noise_beta=0.25, dtype="f32", seed=i), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B); pl.set_cost_spec(env.cost_spec); pl.reset()
        pl.obs0.copy_(torch.as_tensor(0.1*np.random.RandomState(i).randn(17), dtype=pl.dt)); pls.append(pl)
    for _ in range(10): IcemPlanner.plan_step_batch(pls)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): IcemPlanner.plan_step_batch(pls)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B={B}: host enqueue {1e6*t_host/200:.1f} us per step, with GPU {1e6*t_all/200:.1f} us per step", flush=True)
