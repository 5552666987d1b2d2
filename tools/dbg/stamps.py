"""Phase stamps (wall_clock64, 100 MHz) of the merge kernel [0..6] and workgroup 0 of the single-launch kernel [8..14]."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # ICEM_<NAME> variables (incl. ICEM_TILE_ARITH) are mapped per planner: the library reads no environment
env = halfcheetah_env(17)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=ITERS, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
c = env.cost_spec
pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
pl.reset()
obs = 0.1 * np.random.RandomState(0).randn(17)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(5):
    pl.plan_step(obs)
torch.cuda.synchronize()
acc = np.zeros(16)
R = 20
for _ in range(R):
    pl.plan_step(obs); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    acc[:7] += (d[:7] - d[0]) / 100.0
    acc[8:15] += (d[8:15] - d[8]) / 100.0
    gap = (d[0] - d[14]) / 100.0
    acc[15] += gap
acc /= R
print("merge  [us from kernel start]: lists loaded %.2f | keep merged %.2f | threshold+compact %.2f | selected %.2f | gather+refit %.2f | end %.2f" % tuple(acc[1:7]))
# (stamp 12, "wave 0 rolled out", exists only on the Tile16 path; the Tile4 path of small populations goes straight to 13)
print("single [us from kernel start]: staged %.2f | sampled %.2f | tile->HBM issued %.2f | rolled out %.2f | end %.2f"
      % (acc[9], acc[10], acc[11], acc[13], acc[14]))
print("       wave 0 through its 30 steps (stamp 12, in front of the top-K push and the list): %.2f" % acc[12])
print("gap end(single, wg 0) -> start(merge): %.2f us" % acc[15])
print("(with ITERS > 1 the stamps are those of the LAST iteration: the single-launch kernel then carries the previous merge in its prologue"
      " and 'sampled' includes selection + gather + refit + affine map)")
