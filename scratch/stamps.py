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
    dbg = torch.zeros((256*8, 8), dtype=torch.int64, device="cuda")
    L.check(pl.lib.icem_debug_stamps(pl._h, C.c_void_p(dbg.data_ptr())))
    pl.plan_step(obs); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(256, 8, 8)
    w0 = d[:, 0, :]; w0 = w0[w0[:, 0] > 0]
    w7 = d[:, 7, :]; w7 = w7[w7[:, 0] > 0]
    med = lambda x: float(np.median(x))
    print(f"N={N}: WGs={len(w0)}  wave0: sample {med(w0[:,1]-w0[:,0]):.0f}  barrier wait {med(w0[:,2]-w0[:,1]):.0f}  rollout {med(w0[:,3]-w0[:,2]):.0f}  topk+emit {med(w0[:,4]-w0[:,3]):.0f}  total {med(w0[:,4]-w0[:,0]):.0f} cycles")
    print(f"          wave0: RNG+BoxMuller {med(w0[:,5]-w0[:,0]):.0f}  DFT+emit {med(w0[:,1]-w0[:,5]):.0f}")
    print(f"          wave7: sample {med(w7[:,1]-w7[:,0]):.0f}  barrier wait {med(w7[:,2]-w7[:,1]):.0f}")
