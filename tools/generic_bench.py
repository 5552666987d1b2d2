"""Throughput on shapes OUTSIDE the compiled tile-kernel list and in the f64 strict-parity mode -- what a settings/*.json with
another horizon / action count / cost gets.  f64, external noise and ICEM_DISABLE_FAST=1 run the generic kernels
(generic_kernels.hip: any h <= 64, d <= 64, o <= 32, one thread per trajectory); f32 device-noise runs take the exact-f32 GEMM
rollout kernel at any observation width (round 4) and the folded sampler where the horizon is compiled (30, 12, 13, 10).
usage (GPU box): python tools/generic_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)

CASES = [  # (label, h, d, o, N, iters, dtype)
    ("fast path, for reference (h=30 d=6 o=17 f32)", 30, 6, 17, 4096, 5, "f32"),
    ("same shape, f64 strict-parity mode", 30, 6, 17, 4096, 5, "f64"),
    ("h=20 d=8 o=12 f32 (no fast kernel for this shape)", 20, 8, 12, 4096, 5, "f32"),
    ("h=20 d=8 o=12 f32", 20, 8, 12, 65536, 5, "f32"),
    ("h=50 d=3 o=8 f32 (Reacher-sized)", 50, 3, 8, 4096, 5, "f32"),
    ("FetchPickAndPlace (settings/fpp): h=30 d=4 o=28, norm cost", 30, 4, 28, 4096, 5, "f32"),
    ("h=30 d=6 o=17 f32 with ICEM_DISABLE_FAST=1", 30, 6, 17, 65536, 5, "f32"),
]
for label, h, d, o, N, iters, dtype in CASES:
    if "DISABLE_FAST" in label:
        os.environ["ICEM_DISABLE_FAST"] = "1"
    model = DeviceSyntheticModel.make(o, d)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype=dtype, seed=1), -np.ones(d), np.ones(d))
    pl.set_model(model.kind, model.A, model.B)
    if "FetchPickAndPlace" in label:
        from icem_amd.envs import fetch_pick_and_place_env
        pl.set_cost_spec(fetch_pick_and_place_env().cost_spec)   # ||goal - obs[3:6]|| + 0.1 ||obs[0:3] - obs[3:6]|| (robotics.py:150-164)
    else:
        pl.set_cost(0.1, min(8, o - 1), -1.0, 1, 10.0, float(np.pi / 2))
    pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(o), dtype=pl.dt))
    for _ in range(5):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    steps = 50
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.plan_step_resident()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ts = sum(pl.population_sizes) * h
    print(f"{label:58s} N={N:6d}: {dt * 1e6:9.1f} us per MPC step, {ts / dt / 1e9:7.3f} G traj-steps/s", flush=True)
    os.environ.pop("ICEM_DISABLE_FAST", None)
