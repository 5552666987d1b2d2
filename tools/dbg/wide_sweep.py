"""rollout_wide_kernel alone (o = 378, d = 17, h = 30, tanh model) at several populations: us per launch and the f32
matrix-pipe fraction.  One 16-trajectory tile per wavefront, 1 024 wavefront slots on the chip."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402

h, d, o = 30, 17, 378
KIND = int(os.environ.get("WIDE_KIND", "1"))
MODE = int(os.environ.get("WIDE_MODE", "0"))   # icem_set_wide_exact: 0 fp16 planes, 1 exact f32, 2 bf16 planes
model = DeviceSyntheticModel.make(o, d, kind=KIND)
low, high = -0.4 * np.ones(d), 0.4 * np.ones(d)
for n in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384, 32768):
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=n, opt_iters=1, noise_beta=2.0, dtype="f32", seed=1), low, high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost(0.1, 2, -1.0, -1, 0.0, 0.0)
    pl.set_wide_exact(MODE)
    pl.reset()
    obs = 0.1 * np.random.RandomState(0).randn(o)
    acts = (torch.rand(n, h, d, device="cuda") * 0.8 - 0.4).to(pl.dt)
    for _ in range(2):
        pl.rollout_cost(obs, acts)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    reps = 10
    for _ in range(reps):
        pl.rollout_cost(obs, acts)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    fl = 2.0 * (o + d) * o * n * h
    print(f"n={n:6d}: {us:8.1f} us per launch, {fl / us / 1e6:6.1f} TFLOP/s = {100 * fl / us / 1e6 / 157.3:.1f} % of the f32 matrix peak", flush=True)
