"""Phase stamps (wall_clock64, 100 MHz) of iter_ahead_kernel's rollout role, first and last rollout workgroup, thread 0
(needs a library built from tools/experiments/r04_ahead_stamps.patch and ICEM_AHEAD_STAMPS=1).  The stamps are those of the
LAST launch that wrote them = the step's last iteration; pass ITERS to look at other launches (ITERS = 2: launch 1 = 52 428 rows)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
env = halfcheetah_env(17)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=ITERS, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(5):
    pl.plan_step_resident()
torch.cuda.synchronize()
R, acc = 20, np.zeros(16)
for _ in range(R):
    pl.plan_step_resident(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    acc[:8] += (d[:8] - d[0]) / 100.0
    acc[8:] += (d[8:] - d[0]) / 100.0     # the last workgroup on the FIRST one's clock
acc /= R
names = ("entry", "stage 1 + first loads issued", "barrier", "stage 2 + barrier", "gather + refit", "first tile rolled out", "all tiles", "list written")
print(f"N = {N}, {ITERS} iterations, rows of the stamped (last) launch: {pl.population_sizes[-1]}")
for k, n in enumerate(names):
    print(f"  {n:32s} wg 0: {acc[k]:7.2f} us    last rollout wg: {acc[8 + k]:7.2f} us")
