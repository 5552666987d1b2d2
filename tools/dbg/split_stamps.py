import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
from icem_amd import _lib as L
h, d, o = 30, 17, 378
model = DeviceSyntheticModel.make(o, d, kind=1)
low, high = -0.4 * np.ones(d), 0.4 * np.ones(d)
n = 16384
pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=n, opt_iters=1, noise_beta=2.0, dtype="f32", seed=1), low, high)
pl.set_model(model.kind, model.A, model.B); pl.set_cost(0.1, 2, -1.0, -1, 0.0, 0.0); pl.reset()
obs = 0.1 * np.random.RandomState(0).randn(o)
acts = (torch.rand(n, h, d, device="cuda") * 0.8 - 0.4).to(pl.dt)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(3): pl.rollout_cost(obs, acts)
torch.cuda.synchronize()
acc = np.zeros(16)
for _ in range(10):
    pl.rollout_cost(obs, acts); torch.cuda.synchronize()
    v = dbg.cpu().numpy().astype(np.float64); acc += (v[:16] - v[0]) / 100.0
acc /= 10
print("model loop end of waves 0..7 [us from wave 1's step top]:", " ".join("%.2f" % x for x in acc[8:16]))
print("step 5 of wg 3, wave 1 [us]: barrier A %.2f | score %.2f | k-loop %.2f | barrier B %.2f | tanh + state write %.2f | action store %.2f | next step top %.2f" % tuple(acc[1:8]))
