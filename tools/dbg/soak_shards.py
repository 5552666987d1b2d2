"""Randomised soak of the N-sharded path: `world` planners on one GPU (their records concatenated in place of the
RCCL all-gather) must reproduce the single-GPU run bit for bit -- random populations (also smaller than the world:
empty shards), elite counts, iteration counts, flags, seeds, shapes, with and without merge deferral.
usage: soak_shards.py [n_cases] [seed]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402
from icem_amd import _lib as L  # noqa: E402

SHAPES = [(30, 6, 17), (12, 6, 17), (13, 4, 17), (30, 17, 24)]


def np_(t):
    return t.detach().cpu().numpy()


def one(rs, case):
    h, d, o = SHAPES[rs.randint(len(SHAPES))]
    N = int(rs.choice([rs.randint(2, 64), rs.randint(64, 3000), rs.randint(3000, 40000)]))
    if d == 17:
        N = min(N, 12000)
    world = int(rs.choice([2, 3, 4, 8]))
    K, iters, kind = int(rs.randint(2, 12)), int(rs.randint(1, 5)), int(rs.randint(2))
    dtype = "f32" if rs.randint(4) else "f64"
    if dtype == "f64":
        N = min(N, 3000)
    deferral = bool(rs.randint(2))
    kw = dict(horizon=h, act_dim=d, num_traj=N, elites_size=K, opt_iters=iters, dtype=dtype, seed=int(rs.randint(1 << 30)),
              cost_mode=["sum", "best", "final"][rs.randint(3)], noise_beta=float(rs.choice([0.0, 0.25, 2.0])),
              use_mean_actions=bool(rs.randint(2)), keep_previous_elites=bool(rs.randint(2)), shift_elites=bool(rs.randint(2)))
    low, high = -np.ones(d), np.ones(d)
    model = DeviceSyntheticModel.make(o, d, kind=kind)

    def mk(rank, w):
        pl = IcemPlanner(IcemConfig(rank=rank, world=w, **kw), low, high)
        pl.set_model(kind, model.A, model.B)
        pl.set_cost(0.1, 3, -1.0, 1, 10.0, 0.5)
        pl.reset()
        return pl

    single = mk(0, 1)
    pls = [mk(r, world) for r in range(world)]
    st = pls[0]._stream()
    for pl in pls:
        L.check(pl.lib.icem_set_merge_deferral(pl._h, int(deferral)))
    for s in range(2):
        obs = 0.2 * rs.randn(o)
        a1 = np_(single.plan_step(obs))
        for pl in pls:
            pl.obs0.copy_(torch.as_tensor(obs, dtype=pl.dt))
        for it in range(iters):
            for pl in pls:
                L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), s, it, st))
            Kk = pls[0].K
            full = torch.cat([pl.records[r * Kk:(r + 1) * Kk] for r, pl in enumerate(pls)], dim=0)
            for pl in pls:
                pl.records.copy_(full)
                L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), s, it, st))
        ok = all(np.array_equal(np_(pl.executed), a1) and np.array_equal(np_(pl.mean), np_(single.mean))
                 and np.array_equal(np_(pl.std), np_(single.std)) for pl in pls)
        if not ok:
            print("MISMATCH case", case, "step", s, "world", world, "deferral", deferral, kw, "kind", kind, flush=True)
            return False
    return True


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = sum(not one(rs, c) for c in range(n))
    torch.cuda.synchronize()
    print(f"shard soak: {n} random configurations, {bad} mismatching")
    sys.exit(1 if bad else 0)
