"""Phase stamps (wall_clock64, 100 MHz) of one sharded iteration launch on one rank of 8 with its peers absent
(ICEM_XCHG_LOOPBACK): the riding pack workgroup [0..2] and slab workgroup 0 [8..14] of the LAST launch of an MPC step."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["ICEM_XCHG_LOOPBACK"] = "1"
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_gpu = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=per_gpu * world, opt_iters=5, dtype="f32", seed=1, rank=0, world=world),
                 env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
scratch = (C.c_ubyte * L.IPC_HANDLE_BYTES)()
L.check(pl.lib.icem_exchange_create(pl._h, scratch))
L.check(pl.lib.icem_exchange_connect(pl._h, None, None))
pl._exchange = True
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, C.c_void_p(dbg.data_ptr())))
for _ in range(5):
    pl.plan_step_resident()
torch.cuda.synchronize()
R, acc = 30, np.zeros(16)
for _ in range(R):
    pl.plan_step_resident(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    t0 = min(d[0], d[8])
    acc += (d - t0) / 100.0
acc /= R
print("pack workgroup: entry %.2f  selected %.2f  packed+pushed %.2f" % (acc[0], acc[1], acc[2]))
print("slab workgroup 0, selection wave: starts waiting %.2f  flags seen + acquire %.2f  records selected %.2f | all threads: past the barrier %.2f  gathered + refitted %.2f" % (acc[3], acc[7], acc[4], acc[5], acc[6]))
print("slab workgroup 0: entry %.2f  [9] %.2f  [10] %.2f  [11] %.2f  rollout done [12] %.2f  [13] %.2f  end [14] %.2f" % tuple(acc[8:15]))
