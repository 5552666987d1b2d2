"""Stage-wise kernel timings with torch events: rollout on warm (same buffer re-read) and cold (rotating
buffers > MALL) action tensors, and the sampler, for a list of population sizes."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17)
Ns = [int(x) for x in sys.argv[1:]] or [4096, 65536, 131072]
for N in Ns:
    model = DeviceSyntheticModel.make(17, 6)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=1, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    obs = torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=torch.float32, device="cuda")
    mean, std = pl.mean.clone(), pl.std.clone()
    nbuf = max(2, int(600e6 / (N * 720)) + 1)
    bufs = [pl.sample_clip(N, mean, std, offset=i) for i in range(min(nbuf, 64))]
    def timeit(fn, reps=40):
        for i in range(5): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    warm = timeit(lambda i: pl.rollout_cost(obs, bufs[0]))
    cold = timeit(lambda i: pl.rollout_cost(obs, bufs[i % len(bufs)]))
    samp = timeit(lambda i: pl.sample_clip(N, mean, std, offset=i, out=bufs[i % len(bufs)]))
    print(f"N={N}: rollout warm {warm:.1f} us, cold ({len(bufs)} rotating buffers) {cold:.1f} us, sample {samp:.1f} us  (back-to-back launches, includes launch gap)")
