"""What REAL peers cost a sharded rank over absent ones (tools/sharded_rank_bench.py's loopback), as far as one GPU can show
it: `world` processes, each on its own 256 / world CUs (HSA_CU_MASK) with one tile per CU (16 x 256 / world rows per rank),
exchange their records through the library; beside it ONE such rank alone on the same slice with its peers absent
(ICEM_XCHG_LOOPBACK).  The difference is flags and records crossing processes + the ranks' skew -- everything of a node but
the xGMI hop.  usage (GPU box): python tools/dbg/shared_gpu_rank_time.py [world ...]"""
import os, socket, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
STEPS = 300


def planner(rank, world, rows):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=rows * world, opt_iters=5, dtype="f32", seed=1, rank=rank, world=world),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    return pl


def timed(pl, sync=None):
    import torch
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    for _ in range(20):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    if sync:
        sync()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e6


def worker(rank, world, port, rows, out):
    cus = 256 // world
    os.environ["HSA_CU_MASK"] = f"0:{rank * cus}-{(rank + 1) * cus - 1}"
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        independent = out.endswith(".ind")   # the control: the same processes on the same slices, every one an UNSHARDED run of its own
        pl = planner(0 if independent else rank, 1 if independent else world, rows)
        if not independent:
            assert pl.connect_exchange()
        us = timed(pl, dist.barrier)
        t = torch.tensor([us, 0.0 if independent else float(pl.exchange_status()[0])], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            open(out, "w").write(f"{t[0].item():.1f} {int(t[1].item())}")
    finally:
        dist.destroy_process_group()


def alone(world, rows, loopback):
    import ctypes as C, torch
    from icem_amd import _lib as L
    from icem_amd import _lib as _LENV  # noqa: E402
    _LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)
    pl = planner(0, world if loopback else 1, rows)
    if loopback:
        scratch = (C.c_ubyte * L.IPC_HANDLE_BYTES)()
        L.check(pl.lib.icem_exchange_create(pl._h, scratch))
        L.check(pl.lib.icem_exchange_connect(pl._h, None, None))
        pl._exchange = True
    print(f"{timed(pl):.1f}")


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--alone":
        return alone(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "loopback")
    import tempfile
    import torch.multiprocessing as mp
    for world in [int(x) for x in sys.argv[1:]] or [2, 4, 8]:
        cus = 256 // world
        rows = 16 * cus
        env = dict(os.environ, HSA_CU_MASK=f"0:0-{cus - 1}")
        single = float(subprocess.run([sys.executable, __file__, "--alone", str(world), str(rows), "single"], env=env, capture_output=True, text=True, timeout=300).stdout.split()[-1])
        loop = float(subprocess.run([sys.executable, __file__, "--alone", str(world), str(rows), "loopback"], env=dict(env, ICEM_XCHG_LOOPBACK="1"), capture_output=True, text=True,
                                    timeout=300).stdout.split()[-1])
        with tempfile.TemporaryDirectory() as td:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            out = os.path.join(td, "t")
            mp.spawn(worker, args=(world, port, rows, out), nprocs=world, join=True)
            real, status = open(out).read().split()
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            mp.spawn(worker, args=(world, port, rows, out + ".ind"), nprocs=world, join=True)
            ind = open(out + ".ind").read().split()[0]
        print(f"world {world}, {rows} rows per rank on {cus} CUs each: unsharded {rows}-row run on the slice {single:.1f} us per MPC step | one rank of {world}, peers absent "
              f"{loop:.1f} | {world} real processes (slowest rank) {float(real):.1f}, status word {status} | control: {world} processes side by side, each an unsharded "
              f"{rows}-row run (slowest) {float(ind):.1f}", flush=True)


if __name__ == "__main__":
    main()
