import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd import _lib as L
L.set_option("ahead_stamps", 1)
w = bench.WORKLOADS["c2"]; env = bench.make_env(w)
pls = []
for i in range(8):
    model = DeviceSyntheticModel.make(17, 6, seed_a=2*i, seed_b=2*i+1)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=4096, opt_iters=5, noise_beta=0.25, dtype="f32", seed=i), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B); pl.set_cost_spec(env.cost_spec); pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1*np.random.RandomState(i).randn(17), dtype=pl.dt)); pls.append(pl)
for s in range(20):
    IcemPlanner.plan_step_batch(pls)
torch.cuda.synchronize()
print("uploads", pls[0].batch_uploads)
