"""GPU time of ONE rank of a sharded run (its peers absent: ICEM_XCHG_LOOPBACK, every push lands in the rank's own
block and the merges wait for its own flag only): what the local launches + pack-and-push + merges cost per MPC step
through icem_plan_step_sharded -- everything but the wire."""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(10):
    pl.plan_step_resident()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    pl.plan_step_resident()
t_host = (time.perf_counter() - t0) / 100   # enqueue time alone (the stream may still be draining)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
pl.profile_enable(True)
for _ in range(20):
    pl.plan_step_resident()
torch.cuda.synchronize()
prof = pl.profile_read()
print(f"world={world} per-GPU N={per_gpu}: {dt * 1e6:.1f} us per MPC step on this rank (peers absent; host enqueue {t_host * 1e6:.1f} us); "
      + ", ".join(f"{k} {1e3 * v[0] / v[1]:.1f} us x{v[1] // 20}" for k, v in prof.items()), "status", pl.exchange_status())
