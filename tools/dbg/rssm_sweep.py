"""The fused RSSM rollout kernel alone (one launch = a population's 12-step rollout + reward head) at several population
sizes; ICEM_RSSM_TT / ICEM_RSSM_WV pick the tiles per workgroup / waves per workgroup (read per launch / once)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import DeviceRSSMModel  # noqa: E402

h, d = 12, 6
m = DeviceRSSMModel(seed=3)
macs = sum(p.numel() for _, p in m.reference.named_parameters() if p.ndim == 2)
obs = 0.3 * np.random.RandomState(1).randn(230)
for n in [int(a) for a in sys.argv[1:]] or (1024, 16384, 65536):
    acts = torch.rand(n, h, d, device="cuda") * 2 - 1
    for _ in range(3):
        m.rollout_cost(obs, acts)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    reps = 20
    for _ in range(reps):
        m.rollout_cost(obs, acts)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    fl = 2.0 * macs * n * h
    print(f"TT={os.environ.get('ICEM_RSSM_TT', 'auto')} WV={os.environ.get('ICEM_RSSM_WV', '8')} n={n:6d}: {us:8.1f} us, "
          f"{fl / us / 1e6:7.1f} TFLOP/s = {100 * fl / us / 1e6 / 2500:.2f} % of bf16 peak", flush=True)
