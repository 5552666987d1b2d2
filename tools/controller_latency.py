"""End-to-end latency of the drop-in controller: MpcICemHip.get_action (host obs in, host action out) per call."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icem_amd import DeviceSyntheticModel, halfcheetah_env
from icem_amd.controllers import MpcICemHip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = halfcheetah_env(17)
model = DeviceSyntheticModel.make(17, 6)
ctrl = MpcICemHip(env=env, forward_model=model, horizon=30, num_simulated_trajectories=N, factor_decrease_num=1.25,
                  cost_along_trajectory="sum", verbose=False,
                  action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=5, init_std=0.5, use_mean_actions=True,
                                             keep_previous_elites=True, shift_elites_over_time=True,
                                             fraction_elites_reused=0.3, noise_beta=0.25))
obs = 0.1 * np.random.RandomState(0).randn(17)
ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
for _ in range(20):
    a = ctrl.get_action(obs, None)
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    a = ctrl.get_action(obs, None)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"N={N}: get_action median {np.median(ts):.1f} us, p10 {np.percentile(ts, 10):.1f}, p90 {np.percentile(ts, 90):.1f} (action {a[:3]}...)")
