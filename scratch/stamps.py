import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, '.')
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
env = halfcheetah_env(17)
for N in (4096, 65536):
    model = DeviceSyntheticModel.make(17, 6)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=1, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B); c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset(); obs = 0.1*np.random.RandomState(0).randn(17)
    for _ in range(3): pl.plan_step(obs)
    dbg = torch.zeros((2048, 8), dtype=torch.int64, device="cuda")
    L.check(pl.lib.icem_debug_stamps(pl._h, C.c_void_p(dbg.data_ptr())))
    pl.plan_step(obs); torch.cuda.synchronize()
    d = dbg.cpu().numpy(); d = d[d[:, 0] > 0]
    seg = np.stack([d[:,1]-d[:,0], d[:,2]-d[:,1], d[:,3]-d[:,2]], 1)
    print(f"N={N}: waves={len(d)} median cycles: load model+first actions {np.median(seg[:,0]):.0f}, time loop {np.median(seg[:,1]):.0f} ({np.median(seg[:,1])/30:.0f}/step), tile sort {np.median(seg[:,2]):.0f}; loop min/max {seg[:,1].min()}/{seg[:,1].max()}")
    w0 = d[d[:,4] > 0]
    if len(w0): print("   wg merge (wave 0):", np.median(w0[:,4]-w0[:,3]))
