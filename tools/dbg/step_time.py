"""us per MPC step (icem_plan_step, resident inputs) of the c2-shaped workload at the populations given on the command line."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
for N in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384, 32768):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=5, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    for _ in range(10):
        pl.plan_step_resident()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    print(f"N={N}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per MPC step", flush=True)
