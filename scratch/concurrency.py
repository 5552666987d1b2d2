import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17); model = DeviceSyntheticModel.make(17, 6); c = env.cost_spec
for N in (4096, 65536):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=1, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    mean = pl._t(np.zeros((30, 6))); std = pl._t(0.5*np.ones((30, 6))); obs = pl._t(0.1*np.random.RandomState(0).randn(17))
    act = torch.empty((N, 30, 6), device="cuda"); act2 = torch.empty((N, 30, 6), device="cuda")
    pl.sample_clip(N, mean, std, offset=0, out=act); pl.rollout_cost(obs, act); torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def serial(reps):
        for r in range(reps):
            pl.sample_clip(N, mean, std, offset=r, out=act2); pl.rollout_cost(obs, act)
    def concurrent(reps):
        for r in range(reps):
            with torch.cuda.stream(s2): pl.sample_clip(N, mean, std, offset=r, out=act2)
            with torch.cuda.stream(s1): pl.rollout_cost(obs, act)
    for name, fn in (("serial", serial), ("concurrent", concurrent)):
        fn(5); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(50); torch.cuda.synchronize()
        print(f"N={N} {name}: {(time.perf_counter()-t0)/50*1e6:.1f} us per (sample + rollout) pair")
