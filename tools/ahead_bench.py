"""Noise-ahead pipeline against the sampler + rollout pair (GPU box): per population, us per MPC step with
ICEM_NOISE_AHEAD=1 / 0 and whether the two runs agree bit for bit (executed action, mean, std, elites, last pool).
usage: python tools/ahead_bench.py [N ...]   (default 16384 32768 65536 131072)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)


def planner(N, on, iters=5):
    os.environ["ICEM_NOISE_AHEAD"] = "1" if on else "0"   # opt-in; latched by the handle at its first icem_plan_step
    w = dict(bench.WORKLOADS["c4"], N=N, iters=iters)
    pl, _, _ = bench.make_planner(w, 0, 1)
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
    if os.environ.get("ICEM_AB_ONLY"):   # one mode only, for a kernel trace: ICEM_AB_ONLY=ahead|pair
        for N in sizes:
            pl = planner(N, os.environ["ICEM_AB_ONLY"] == "ahead")
            print(f"N={N:7d}  {os.environ['ICEM_AB_ONLY']} {timed(pl, 100):8.1f} us/step", flush=True)
        return
    for N in sizes:
        a, b = planner(N, True), planner(N, False)
        # same number of steps so far on both: compare the state after a few more
        same = True
        for _ in range(3):
            a.plan_step_resident()
            b.plan_step_resident()
            torch.cuda.synchronize()
            n_last = a.population_sizes[-1]
            same = same and torch.equal(a.executed, b.executed) and torch.equal(a.mean, b.mean) and torch.equal(a.std, b.std) \
                and torch.equal(a.current_elites()[0], b.current_elites()[0]) and torch.equal(a.current_elites()[1], b.current_elites()[1]) \
                and torch.equal(a.actions[:n_last], b.actions[:n_last]) and torch.equal(a.costs[:n_last], b.costs[:n_last])
        ta, tb = timed(a), timed(b)
        print(f"N={N:7d}  ahead {ta:8.1f} us/step   sampler+rollout {tb:8.1f} us/step   ratio {tb / ta:5.2f}   bit-equal {same}", flush=True)


if __name__ == "__main__":
    main()
