"""The in-library exchange (IPC-mapped peer blocks + flags, csrc/exchange.hip) between `world` PROCESSES that share the one
GPU of a box, at a global population small enough that every rank's workgroups are resident together (a rank's spinning wait
cannot keep a peer's pack workgroup off the chip): every rank's actions, mean and elites of a few MPC steps against the
single-process run, bit for bit, and the path each rank ended on.  ICEM_SHARED_SCALE=<x> scales the populations,
ICEM_SHARED_SLICES=1 gives every rank its own slice of the CUs (HSA_CU_MASK) -- larger populations then fit side by side.  (At bench.py's populations 4+ ranks on one GPU do not fit
together: the bounded waits run out and the ranks step down -- profiles/r05_exchange_fault_drills.txt.)
usage (GPU box): python tools/dbg/shared_gpu_worlds.py [world ...]"""
import os, socket, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SCALE = float(os.environ.get("ICEM_SHARED_SCALE", "1"))   # global populations x SCALE (to find where the ranks stop fitting together)
SHAPES = {"halfcheetah": dict(N=int(2000 * SCALE), iters=3), "door": dict(N=int(1500 * SCALE), iters=3)}


def make(name, rank, world):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    from icem_amd import envs as E
    env = halfcheetah_env(17) if name == "halfcheetah" else E.door_env()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=1)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=d, num_traj=SHAPES[name]["N"], opt_iters=SHAPES[name]["iters"], dtype="f32", seed=21,
                                noise_beta=0.25 if name == "halfcheetah" else 2.5, rank=rank, world=world), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    return pl, o


def run(pl, o, steps=4):
    pl.reset()
    out = []
    for s in range(steps):
        a = pl.plan_step(0.1 * np.random.RandomState(s).randn(o)).cpu().numpy().copy()
        ea, ec = pl.current_elites()
        out += [a, pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy(), ea.cpu().numpy().copy(), ec.cpu().numpy().copy()]
    return out


def worker(rank, world, port, out_dir):
    if os.environ.get("ICEM_SHARED_SLICES"):   # every rank its own slice of the 256 CUs (before HSA starts in this process)
        cus = 256 // world
        os.environ["HSA_CU_MASK"] = f"0:{rank * cus}-{(rank + 1) * cus - 1}"
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for name in SHAPES:
            pl, o = make(name, rank, world)
            ok = pl.connect_exchange()
            try:
                res = run(pl, o)
            except Exception as e:   # a bounded wait ran out: said below through the status word
                res = [np.zeros(1)] * 20
            status = pl.exchange_status()[0] if ok else -1
            np.savez(os.path.join(out_dir, f"{name}_r{rank}.npz"), *res, ok=np.array([int(ok), status]))
            dist.barrier()
    finally:
        dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    worlds = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
    single = {}
    for name in SHAPES:
        pl, o = make(name, 0, 1)
        single[name] = run(pl, o)
    bad = 0
    for world in worlds:
        with tempfile.TemporaryDirectory() as td:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            mp.spawn(worker, args=(world, port, td), nprocs=world, join=True)
            for name in SHAPES:
                same, paths = True, []
                for r in range(world):
                    z = np.load(os.path.join(td, f"{name}_r{r}.npz"))
                    got = [z[f"arr_{i}"] for i in range(len(single[name]))]
                    same &= all(np.array_equal(u, v) for u, v in zip(got, single[name]))
                    paths.append(tuple(int(v) for v in z["ok"]))
                bad += not same
                print(f"world {world} {name:11s}: every rank {'identical to' if same else 'DIFFERS from'} the single-process run; (exchange connected, status word) per rank: {paths}", flush=True)
    print("shared-GPU worlds:", "all identical" if not bad else f"{bad} MISMATCHING")


if __name__ == "__main__":
    main()
