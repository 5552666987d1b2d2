"""c2 in float64 (the strict-parity mode) for a kernel trace: 30 MPC steps of N = 4096 x 5 iterations through icem_plan_step.
usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python tools/dbg/f64_profile.py [N] [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=5, dtype="f64", seed=1234), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
for _ in range(4):
    pl.plan_step_resident()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pl.plan_step_resident()
torch.cuda.synchronize()
print(f"f64 N={N}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per MPC step")
