#!/usr/bin/env python3
"""Wall time of the REFERENCE's own MpcICem.get_action (imported from /root/reference with the same stand-in modules
as make_golden.py) on the benchmark's synthetic workloads -- build container only (the reference does not travel);
the result is committed as profiles/rNN_reference_cpu_walltime.json, a recorded number next to bench.py's live
cpu_baseline.  Single thread like the reference (main.py:23 forces OMP_NUM_THREADS=1).

usage: python tests/golden/time_reference.py profiles/r01_reference_cpu_walltime.json
"""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as M


def run(N, h, d, o, beta, iters, steps):
    from controllers.icem import MpcICem
    from models.abstract_models import ForwardModelWithDefaults
    from gym import spaces
    import environments.mujoco as ref_mj
    A, B = M.make_model_mats(o, d)

    class Env:
        name = "FakeHalfCheetah"
        action_space = spaces.Box(low=-np.ones(d), high=np.ones(d))
        penalise_flipping = True

        def cost_fn(self, obs, act, next_obs):
            return ref_mj.HalfCheetahMaybeWithPosition.cost_fn(self, obs, act, next_obs)

    class Model(ForwardModelWithDefaults):
        def train(self, buffer): pass
        def save(self, path): pass
        def load(self, path): pass

        def predict(self, *, observations, states, actions):   # one batched matmul per step: the cheapest model there is
            return observations @ A + actions @ B, None, np.zeros(observations.shape[:-1] + (1,))

    env = Env()
    ctrl = MpcICem(env=env, forward_model=Model(env=env), horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                   cost_along_trajectory="sum", verbose=False, do_visualize_plan=False,
                   action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5,
                                              use_mean_actions=True, keep_previous_elites=True,
                                              shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=beta))
    np.random.seed(0)
    obs = 0.1 * np.random.RandomState(0).randn(o)
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctrl.get_action(obs, None)
        times.append(time.perf_counter() - t0)
    pops, n = [], N
    for i in range(iters):
        if i:
            n = max(2 * 10, int(n / 1.25))
        pops.append(n)
    ts = sum(pops) * h
    best = min(times)
    return {"N": N, "h": h, "d": d, "o": o, "beta": beta, "iters": iters, "mpc_steps_timed": steps,
            "seconds_per_mpc_step": times, "best_seconds_per_mpc_step": best, "traj_steps_per_mpc_step": ts,
            "traj_steps_per_s": ts / best}


def main():
    if not os.path.isdir(M.REF):
        sys.exit("needs /root/reference (build container only)")
    M.install_stubs()
    cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    out = {"what": "reference MpcICem.get_action (icem/controllers/icem.py:106-189) imported from /root/reference, synthetic "
                   "linear model + the reference's HalfCheetah cost_fn, single thread",
           "host": {"cpu_model": cpu_model, "cpus": os.cpu_count(), "python": sys.version.split()[0], "numpy": np.__version__},
           "c1": run(128, 30, 6, 17, 0.25, 3, 3), "c2": run(4096, 30, 6, 17, 0.25, 5, 2)}
    path = sys.argv[1] if len(sys.argv) > 1 else "reference_cpu_walltime.json"
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: (v["best_seconds_per_mpc_step"], v["traj_steps_per_s"]) for k, v in out.items() if k in ("c1", "c2")}))


if __name__ == "__main__":
    main()
