"""Per-kernel-class times of the FetchPickAndPlace-shaped workload (h = 30, d = 4, o = 28, the env's norm cost: settings/fpp,
environments/robotics.py:150-164) -- a narrow model the tile kernels do not serve, rolled out by the exact-f32 GEMM kernel."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd.envs import fetch_pick_and_place_env
h, d, o, N = 30, 4, 28, 4096
model = DeviceSyntheticModel.make(o, d)
pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=5, dtype="f32", seed=1), -np.ones(d), np.ones(d))
pl.set_model(model.kind, model.A, model.B)
spec = fetch_pick_and_place_env().cost_spec
print("spec:", spec)
pl.set_cost_spec(spec)
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(o), dtype=pl.dt))
for _ in range(5): pl.plan_step_resident()
torch.cuda.synchronize()
pl.profile_enable(True)
for _ in range(10): pl.plan_step_resident()
torch.cuda.synchronize()
print({k: (round(1e3 * v[0] / v[1], 1), v[1] // 10) for k, v in pl.profile_read().items()})

