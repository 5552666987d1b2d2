import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17); model = DeviceSyntheticModel.make(17, 6); c = env.cost_spec
for N in (4096, 8192, 16384, 32768, 49152, 65536, 131072):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=1, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    act = pl.sample_clip(N, np.zeros((30, 6)), 0.5*np.ones((30, 6)), offset=1)
    obs = 0.1*np.random.RandomState(0).randn(17)
    for _ in range(3): pl.rollout_cost(obs, act)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    o = pl._t(obs)
    for rep in range(10):
        ev[0].record(); pl.rollout_cost(o, act); ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1])*1e3)
    ts2 = []
    for rep in range(10):
        ev[0].record(); pl.sample_clip(N, pl.mean if hasattr(pl,'mean') else np.zeros((30,6)), 0.5*np.ones((30,6)), offset=rep, out=act); ev[1].record(); torch.cuda.synchronize(); ts2.append(ev[0].elapsed_time(ev[1])*1e3)
    print(f"N={N:7d}: rollout us min {min(ts):7.1f} med {np.median(ts):7.1f} max {max(ts):7.1f} | sample us min {min(ts2):7.1f} med {np.median(ts2):7.1f}")
