import sys, numpy as np, torch
sys.path.insert(0, '.')
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17); model = DeviceSyntheticModel.make(17, 6); c = env.cost_spec
N = 65536
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=1, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
mean = pl._t(np.zeros((30, 6))); std = pl._t(0.5*np.ones((30, 6))); obs = pl._t(0.1*np.random.RandomState(0).randn(17))
act = torch.empty((N, 30, 6), device="cuda")
pl.profile_enable(True)
for rep in range(6):
    pl.sample_clip(N, mean, std, offset=rep, out=act)
    pl.rollout_cost(obs, act)
    pl.rollout_cost(obs, act)
torch.cuda.synchronize()
import ctypes as C
from icem_amd import _lib as L
# read individual spans: use profile_read totals only -> do per-kind via separate loops
print(pl.profile_read())
for rep in range(6):
    pl.sample_clip(N, mean, std, offset=rep, out=act)
    torch.cuda.synchronize()
    pl.profile_read()
    pl.rollout_cost(obs, act); torch.cuda.synchronize(); a = pl.profile_read()["rollout_cost"][0]
    pl.rollout_cost(obs, act); torch.cuda.synchronize(); b = pl.profile_read()["rollout_cost"][0]
    print(f"after sample+sync: first rollout {a*1e3:.1f} us, second {b*1e3:.1f} us")
