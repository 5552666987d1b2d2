"""Would replaying an MPC step as a HIP graph shorten it?  Captures two icem_plan_step calls (the ping-pong buffers come back
after two) and replays them -- with stale RNG offsets, so a timing probe only -- against the same steps enqueued on a stream."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env
env = halfcheetah_env(17); model = DeviceSyntheticModel.make(17, 6)
for N in (4096, 65536):
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=5, dtype="f32", seed=1), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B); pl.set_cost_spec(env.cost_spec); pl.reset()
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    for _ in range(10): pl.plan_step_resident()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): pl.plan_step_resident()
    torch.cuda.synchronize(); print(N, "stream: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): pl.plan_step_resident()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            pl.plan_step_resident(); pl.plan_step_resident()   # two steps: the ping-pong buffers come back
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): g.replay()
        torch.cuda.synchronize(); print(N, "graph (2 steps per replay, stale RNG offsets): %.1f us per step" % ((time.perf_counter() - t0) / 200 * 1e6))
    except Exception as e:
        print("capture failed:", repr(e)[:300])
