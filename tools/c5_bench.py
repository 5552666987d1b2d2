"""BASELINE configs[4] (N=1024, h=12, d=6, beta=0.25, learned dynamics): MpcICemHip.get_action with the declared RSSM
(icem_amd.models.declared_rssm) in f32 and bf16 -- ms per MPC step, and the model's GEMM flops over that time
against the dense bf16 matrix-core peak (2.5 PFLOP/s).  The model steps are torch / hipBLASLt launches: at this size
the step is launch bound, which is what the number shows."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from icem_amd import DeviceRSSMModel, MpcICemHip, declared_rssm, halfcheetah_env  # noqa: E402

N, h, d, iters = 1024, 12, 6, 5
env = halfcheetah_env(17)
asp = dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5, use_mean_actions=True, keep_previous_elites=True,
           shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=0.25)
for dtype in (None, torch.bfloat16, "fused"):
    m = DeviceRSSMModel(seed=3) if dtype == "fused" else declared_rssm(seed=3, dtype=dtype)
    ctrl = MpcICemHip(env=env, forward_model=m, horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                      cost_along_trajectory="sum", dtype="f32", seed=1, action_sampler_params=asp)
    obs = 0.3 * np.random.RandomState(1).randn(230)
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    for _ in range(5):
        ctrl.get_action(obs, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 30
    for _ in range(reps):
        ctrl.get_action(obs, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    net = m.reference if dtype == "fused" else m.module
    macs = sum(p.numel() for n_, p in net.named_parameters() if p.ndim == 2)   # one MAC per weight per trajectory-step
    pops = ctrl.planner.population_sizes
    flops = 2.0 * macs * sum(pops) * h
    label = {None: "torch f32 (graph)  ", torch.bfloat16: "torch bf16 (graph) ", "fused": "fused HIP bf16 MFMA"}[dtype]
    print(f"{label}: {dt * 1e3:7.3f} ms per MPC step ({sum(pops)} trajectories x h={h}); "
          f"{flops / dt / 1e12:6.2f} TFLOP/s of model GEMMs = {100 * flops / dt / 2.5e15:.3f} % of the dense bf16 peak; "
          f"{sum(pops) * h / dt / 1e6:.2f} M traj-steps/s")

# the fused kernel alone: one launch = a whole population's 12-step rollout + reward head
m = DeviceRSSMModel(seed=3)
macs = sum(p.numel() for _, p in m.reference.named_parameters() if p.ndim == 2)
obs = 0.3 * np.random.RandomState(1).randn(230)
for n in (1024, 4096, 16384, 65536):
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
    print(f"icem_rssm_rollout_cost n={n:6d}: {us:8.1f} us per launch, {fl / us / 1e6:7.1f} TFLOP/s = {100 * fl / us / 1e6 / 2500:.2f} % of the "
          f"dense bf16 peak, {n * h / us:.1f} M traj-steps/s")
