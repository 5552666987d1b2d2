"""Phase stamps (wall_clock64, 100 MHz) of the one-launch step of small populations (k_step_xcd.hip): members 0 and 31, every
iteration -- [0] top of the iteration, [1] raw noise requested, [2] barrier passed + selection done, [3] gather + refit done,
[4] mapped, [5] rolled out, [6] list emitted, [7] arrived; then the epilogue.  (option ahead_stamps = 1 + icem_debug_stamps)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L.set_option("ahead_stamps", 1)
L.set_option("step_xcd", 1)   # (off by default: the path under study here)
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=ITERS, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
pl.set_model(model.kind, model.A, model.B)
pl.set_cost_spec(env.cost_spec)
pl.reset()
obs = 0.1 * np.random.RandomState(0).randn(17)
dbg = torch.zeros(16 + 256, dtype=torch.int64, device="cuda")
L.check(pl.lib.icem_debug_stamps(pl._h, dbg.data_ptr()))
for _ in range(5):
    pl.plan_step(obs)
torch.cuda.synchronize()
R = 20
acc = np.zeros((2, ITERS + 1, 8))
for _ in range(R):
    pl.plan_step(obs); torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64)
    t0 = d[0]
    for m in range(2):
        blk = d[16 + 128 * m: 16 + 128 * m + 8 * (ITERS + 1)].reshape(ITERS + 1, 8)
        acc[m] += (blk - t0) / 100.0
acc /= R
print("status", pl.step_status())
names = ["top", "copy waves out", "sel done", "refit done", "mapped", "last rollout wave out", "emitted", "arrived"]
for m in range(2):
    print(f"member {0 if m == 0 else 31}: us from member 0's entry")
    for it in range(ITERS):
        print(f"  it {it}: " + " | ".join(f"{n} {acc[m, it, k]:.2f}" for k, n in enumerate(names)))
    print(f"  epilogue done {acc[m, ITERS, 0]:.2f} | left {acc[m, ITERS, 1]:.2f}")
