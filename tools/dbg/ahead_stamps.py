"""Phase stamps (wall_clock64, 100 MHz) of iter_ahead_kernel's rollout role, first and last rollout workgroup, thread 0, for
every iteration's launch of an MPC step (ICEM_AHEAD_STAMPS=1 is set here; ICEM_TILE_ARITH=0|1 picks the tile arithmetic).
usage: python tools/dbg/ahead_stamps.py [N [ITERS]]"""
import sys, os, numpy as np, torch
os.environ.setdefault("ICEM_AHEAD_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)
env = halfcheetah_env(17)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=ITERS, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
dbg = torch.zeros(16 + 32 * ITERS, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(5):
    pl.plan_step_resident()
torch.cuda.synchronize()
R, acc = 20, np.zeros((ITERS, 32))
for _ in range(R):
    pl.plan_step_resident(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)[16:].reshape(ITERS, 32)
    acc += (d - d[0, 0]) / 100.0      # everything on the clock of the step's first stamp
acc /= R
names = ("entry", "stage 1, first loads issued", "barrier", "selection + barrier", "gather + refit", "first tile", "all tiles", "list written")
print(f"N = {N}, {ITERS} iterations, tile arithmetic {pl.tile_arith}; us since the step's first stamp (first rollout workgroup | last); rows per launch {pl.population_sizes}")
print("  launch  " + "  ".join(f"{n[:18]:>18s}" for n in names))
for it in range(ITERS):
    for wg, off in (("first", 0), ("last", 8)):
        print(f"  {it} {wg:5s} " + "  ".join(f"{acc[it, off + k]:18.2f}" for k in range(8)))
for it in range(ITERS):
    a = acc[it]
    end = max(a[7], a[15])
    nxt = acc[it + 1, 0] if it + 1 < ITERS else float("nan")
    print(f"  launch {it}: shift wg {a[16]:7.2f} .. {a[17]:7.2f}   first noise wg {a[20]:7.2f} (sampled {a[22]:7.2f}, barrier {a[23]:7.2f}) .. {a[21]:7.2f}   last noise wg {a[24]:7.2f} ({a[26]:7.2f}, {a[27]:7.2f}) .. {a[25]:7.2f}")
    print(f"  launch {it}: entry -> tiles start {a[12] - a[8]:5.2f} (last wg), tiles {a[14] - a[12]:6.2f}, list {a[15] - a[14]:5.2f}, end -> next entry {nxt - end:5.2f}")
