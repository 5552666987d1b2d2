import sys; sys.path.insert(0, ".")
import numpy as np, torch
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
env = halfcheetah_env(17); model = DeviceSyntheticModel.make(17, 6, kind=0)
for dtype in ("f32", "f64"):
    for N in (4096, 300):
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=3, dtype=dtype, seed=1), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B); pl.set_cost_spec(env.cost_spec); pl.reset()
        ob = 0.1 * np.random.RandomState(0).randn(17); ob[8] = np.nan
        a = pl.plan_step(ob); torch.cuda.synchronize()
        print(dtype, N, "executed", a.cpu().numpy(), "mean finite", bool(torch.isfinite(pl.mean).all()), "best_cost", pl.best_cost.cpu().numpy(),
              "elite costs", pl.elites_costs[0, :3].cpu().numpy())
