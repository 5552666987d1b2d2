"""Phase stamps of select_refit_kernel (the strict-parity path's one-launch selection + gather + refit), us from its entry.
usage (GPU box): python tools/dbg/f64_stamps.py [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
from icem_amd import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=5, dtype="f64", seed=1234), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
obs = 0.1 * np.random.RandomState(0).randn(17)
dbg = torch.zeros(32, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(3):
    pl.plan_step(obs)
acc = np.zeros(6)
acr = np.zeros(5)
R = 20
for _ in range(R):
    pl.plan_step(obs)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    acc += (d[17:23] - d[16]) / 100.0
    acr += (d[25:30] - d[24]) / 100.0
acc /= R
acr /= R
print('rollout_cost_rows (workgroup 0, last iteration) [us from entry]: operands + barrier %.2f | actions staged, step 0 starts %.2f | step 1 starts %.2f | step 2 starts %.2f | end %.2f' % tuple(acr))
print("select_refit (last iteration) [us from entry]: bests found %.2f | wave thresholds %.2f | candidates collected %.2f | placed %.2f | gathered + refitted %.2f   (candidates: %d)"
      % (acc[0], acc[1], acc[2], acc[3], acc[4], int(d[23])))
