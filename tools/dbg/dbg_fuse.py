import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17)
N = int(sys.argv[1]); iters = int(sys.argv[2])
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=iters, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
c = env.cost_spec
pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
pl.reset()
obs = 0.1*np.random.RandomState(0).randn(17)
print("pops", pl.population_sizes, flush=True)
for s in range(3):
    pl.plan_step(obs); torch.cuda.synchronize(); print("step", s, "ok", float(pl.costs[:8].sum()), flush=True)
