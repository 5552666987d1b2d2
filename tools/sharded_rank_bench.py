"""GPU time of ONE rank of a sharded run (no communication: the other ranks' records stay as they are): what the
local launches + pack (+ the last merge) cost per MPC step, to be added to the all-gather latency."""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
from icem_amd import _lib as L
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_gpu = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
for deferral in (0, 1):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=per_gpu * world, opt_iters=5, dtype="f32", seed=1, rank=0, world=world),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    L.check(pl.lib.icem_set_merge_deferral(pl._h, deferral))
    st = pl._stream()
    def step(s):
        for it in range(5):
            L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), s, it, st))
            L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), s, it, st))
    for s in range(10): step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(10, 110): step(s)
    t_host = (time.perf_counter() - t0) / 100   # enqueue time alone (the stream may still be draining)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    pl.profile_enable(True)
    for s in range(110, 130): step(s)
    torch.cuda.synchronize()
    prof = pl.profile_read()
    print(f"world={world} per-GPU N={per_gpu} deferral={deferral}: {dt * 1e6:.1f} us per MPC step on this rank (no all-gather; host enqueue {t_host * 1e6:.1f} us); "
          + ", ".join(f"{k} {1e3 * v[0] / v[1]:.1f} us x{v[1] // 20}" for k, v in prof.items()))
