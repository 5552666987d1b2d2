"""The fp16-plane tile (icem_set_tile_arith 1, Tile16H) against the exact f32 tile (0) on the GPU box: per population, us per
MPC step in both arithmetics on the same box, the largest relative cost difference over the last pool (scaled by the
costs' magnitude), and whether the two runs select the same elites.
usage: python tools/tile_arith_bench.py [N ...]   (default 16384 32768 65536 131072)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def planner(N, arith, iters=5, wl="c4"):
    w = dict(bench.WORKLOADS[wl], N=N, iters=iters)
    pl, _, _ = bench.make_planner(w, 0, 1)
    got = pl.set_tile_arith(arith)
    assert got == arith, f"tile arithmetic {arith} not served (got {got})"
    pl.plan_step_resident()
    return pl


def timed(pl, steps=200):
    for _ in range(20):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768, 65536, 131072]
    for N in sizes:
        a, b = planner(N, 1), planner(N, 0)
        torch.cuda.synchronize()
        n_last = a.population_sizes[-1]
        # step 0 ran on both from the same state: same noise, and the same actions as long as the elites agree
        same_act = torch.equal(a.actions[:n_last], b.actions[:n_last])
        ca, cb = a.costs[:n_last].double(), b.costs[:n_last].double()
        err = float((ca - cb).abs().max() / cb.abs().max())
        el = torch.equal(a.current_elites()[0], b.current_elites()[0])
        ta, tb = timed(a), timed(b)
        ta2, tb2 = timed(a), timed(b)
        print(f"N={N:7d}  f16x2 {min(ta, ta2):8.1f} us/step   f32 {min(tb, tb2):8.1f} us/step   ratio {min(tb, tb2) / min(ta, ta2):5.2f}   "
              f"same last pool {same_act}   max |dcost| / max |cost| {err:.2e}   same elites {el}", flush=True)


if __name__ == "__main__":
    main()
