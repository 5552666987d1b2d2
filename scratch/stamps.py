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
    dbg = torch.zeros((1024, 8), dtype=torch.int64, device="cuda")
    L.check(pl.lib.icem_debug_stamps(pl._h, C.c_void_p(dbg.data_ptr())))
    pl.plan_step(obs); torch.cuda.synchronize()
    d = dbg.cpu().numpy(); d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    names = ["start->S done", "S->W done(R start)", "R", "K", "total(first tile)"]
    seg = np.stack([d[:,1]-d[:,0], d[:,2]-d[:,1], d[:,3]-d[:,2], d[:,4]-d[:,3], d[:,5]-d[:,0]], 1)
    print(f"N={N}: WGs={len(d)}  cycles median per phase (first tile of each WG):")
    for i, nme in enumerate(names): print(f"   {nme:24s} median {np.median(seg[:,i]):9.0f}  min {seg[:,i].min():9.0f} max {seg[:,i].max():9.0f}")
    print("   kernel span cycles:", d[:,5].max() - t0, " start skew:", d[:,0].max()-t0)
