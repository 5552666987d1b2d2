#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE's own ``MpcICem``.

Runs only in the build container (needs ``/root/reference``); the reference's
source never travels -- only the ``.npz`` data this script writes (inputs and
expected outputs) is committed under ``tests/golden/``.

The reference imports four modules that are absent from the image
(``allogger``, ``forwardable``, ``gym``, ``colorednoise``) plus, for its cost
functions, ``gym.envs.mujoco`` / ``mujoco_py``.  Minimal stand-ins are written
to a temp directory at run time:

* ``allogger`` / ``forwardable`` / ``gym`` / ``mujoco_py``: interface stubs only
  (no arithmetic).
* ``colorednoise``: third-party, unpinned, not vendored by the reference.  The
  stand-in restates the published 1.x ``powerlaw_psd_gaussian`` using the
  *module-level* ``numpy.random.normal`` so it consumes the global legacy
  stream exactly as upstream does.

What is captured (per case): the white draws of every sampling call (recovered
by replaying the saved ``np.random`` state), the simulated action batches,
pool costs, elite indices, refit mean/std, best index and executed action of
every CEM iteration of several consecutive MPC steps, the fake-model matrices,
and direct outputs of the reference's HalfCheetah / HumanoidStandup
``cost_fn``.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import textwrap
import types

import numpy as np

REF = "/root/reference/icem"
OUT = os.path.dirname(os.path.abspath(__file__))

STUBS = {
    "allogger.py": """
        class _L:
            logdir = "/tmp"
            def log(self, *a, **k): pass
            def info(self, *a, **k): pass
        def get_logger(*a, **k): return _L()
    """,
    "forwardable.py": """
        import sys
        def forwardable():
            return lambda cls: cls
        def def_delegators(attr, names):
            ns = sys._getframe(1).f_locals
            for name in [n.strip() for n in names.split(",")]:
                def mk(name):
                    def f(self, *a, **k):
                        return getattr(getattr(self, attr), name)(*a, **k)
                    f.__name__ = name
                    return f
                ns[name] = mk(name)
    """,
    "gym/__init__.py": """
        from . import spaces
        class Env:
            def __init__(self, *a, **k): pass
    """,
    "gym/spaces.py": """
        import numpy as np
        class Space: pass
        SAMPLE_LOG = []
        class Box(Space):
            def __init__(self, low, high, dtype=np.float32):
                self.low = np.asarray(low, dtype=dtype); self.high = np.asarray(high, dtype=dtype)
                self.shape = self.low.shape
            def sample(self):   # bounded Box.sample: uniform(low, high); the unit draws are logged for replay
                u = np.random.random_sample(self.shape)
                SAMPLE_LOG.append(u)
                return (self.high.astype(np.float64) - self.low) * u + self.low
        class Discrete(Space):
            def __init__(self, n): self.n = n
        class Dict(Space):
            def __init__(self, d): self.spaces = d
    """,
    "gym/utils.py": """
        class EzPickle:
            def __init__(self, *a, **k): pass
    """,
    "gym/envs/__init__.py": "",
    "gym/envs/mujoco/__init__.py": """
        class MujocoEnv:
            def __init__(self, *a, **k): pass
        class ReacherEnv(MujocoEnv): pass
    """,
    "gym/envs/mujoco/half_cheetah_v3.py": "from . import MujocoEnv\nclass HalfCheetahEnv(MujocoEnv): pass\n",
    "gym/envs/mujoco/ant_v3.py": "from . import MujocoEnv\nclass AntEnv(MujocoEnv): pass\n",
    "gym/envs/mujoco/humanoid_v3.py": "from . import MujocoEnv\nclass HumanoidEnv(MujocoEnv): pass\n",
    "gym/envs/mujoco/humanoidstandup.py": "from . import MujocoEnv\nclass HumanoidStandupEnv(MujocoEnv): pass\n",
    "gym/envs/mujoco/hopper_v3.py": "from . import MujocoEnv\nclass HopperEnv(MujocoEnv): pass\n",
    "gym/envs/robotics/__init__.py": "",
    "gym/envs/robotics/robot_env.py": "class RobotEnv:\n    def __init__(self, *a, **k): pass\n",
    "gym/envs/robotics/fetch/__init__.py": "",
    "gym/envs/robotics/fetch/pick_and_place.py": "class FetchPickAndPlaceEnv:\n    def __init__(self, *a, **k): pass\n",
    "gym/envs/robotics/fetch/reach.py": "class FetchReachEnv:\n    def __init__(self, *a, **k): pass\n",
    "mujoco_py/__init__.py": "",
    "mujoco_py/generated/__init__.py": "",
    "mujoco_py/generated/const.py": "CAMERA_FIXED = 2\n",
    # restatement of colorednoise 1.x (third-party, unpinned: Pipfile:10)
    "colorednoise.py": """
        from numpy import sqrt, newaxis
        from numpy.fft import irfft, rfftfreq
        from numpy.random import normal
        from numpy import sum as npsum
        CALLS = []
        def powerlaw_psd_gaussian(exponent, size, fmin=0):
            import numpy as np
            state = np.random.get_state()
            try:
                size = list(size)
            except TypeError:
                size = [size]
            samples = size[-1]
            f = rfftfreq(samples)
            s_scale = f
            fmin = max(fmin, 1. / samples)
            ix = npsum(s_scale < fmin)
            if ix and ix < len(s_scale):
                s_scale[:ix] = s_scale[ix]
            s_scale = s_scale ** (-exponent / 2.)
            w = s_scale[1:].copy()
            w[-1] *= (1 + (samples % 2)) / 2.
            sigma = 2 * sqrt(npsum(w ** 2)) / samples
            size[-1] = len(f)
            dims_to_add = len(size) - 1
            s_scale = s_scale[(newaxis,) * dims_to_add + (Ellipsis,)]
            sr = normal(scale=s_scale, size=size)
            si = normal(scale=s_scale, size=size)
            after = np.random.get_state()
            # recover the underlying standard-normal draws by replaying the stream
            np.random.set_state(state)
            z_r = np.random.normal(size=size)
            z_i = np.random.normal(size=size)
            assert np.array_equal(z_r * s_scale, sr) and np.array_equal(z_i * s_scale, si)
            a2 = np.random.get_state()
            assert a2[2] == after[2] and np.array_equal(a2[1], after[1])
            if not (samples % 2):
                si[..., -1] = 0
            si[..., 0] = 0
            s = sr + 1J * si
            y = irfft(s, n=samples, axis=-1) / sigma
            CALLS.append((z_r, z_i, y))
            return y
    """,
}


def install_stubs():
    root = tempfile.mkdtemp(prefix="icem_stubs_")
    for rel, body in STUBS.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(body))
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, root)
    return root


def make_model_mats(o, d, seed_a=0, seed_b=1):
    A = 0.95 * np.eye(o) + 0.05 * np.random.RandomState(seed_a).randn(o, o) / np.sqrt(o)
    B = 0.1 * np.random.RandomState(seed_b).randn(d, o)
    return A, B


def run_case(name, *, N, h, d, o, beta, iters, seed, n_steps, kind, env_kind,
             cost_mode="sum", K=10, xi=0.3, gamma=1.25, alpha=0.1, init_std=0.5,
             use_mean=True, keep=True, shift=True, bounds=1.0, new_mean=False):
    import colorednoise
    from controllers.icem import MpcICem as RefMpcICem
    from models.abstract_models import ForwardModelWithDefaults
    from gym import spaces
    import environments.mujoco as ref_mj

    A, B = make_model_mats(o, d)

    class FakeEnv:
        def __init__(self):
            self.name = "Fake" + env_kind
            self.action_space = spaces.Box(low=-bounds * np.ones(d), high=bounds * np.ones(d))
            self.penalise_flipping = True

        def cost_fn(self, obs, act, next_obs):
            # the REFERENCE's own cost functions, called unbound on this object
            if env_kind == "halfcheetah":
                return ref_mj.HalfCheetahMaybeWithPosition.cost_fn(self, obs, act, next_obs)
            return ref_mj.HumanoidStandup.cost_fn(self, obs, act, next_obs)

        def reward_fn(self, obs, act, next_obs):
            return -self.cost_fn(obs, act, next_obs)

    class FakeModel(ForwardModelWithDefaults):
        def train(self, buffer): pass
        def save(self, path): pass
        def load(self, path): pass

        def predict(self, *, observations, states, actions):
            nxt = np.zeros_like(observations)
            for k in range(o):
                nxt = nxt + observations[..., k:k + 1] * A[k]
            for j in range(d):
                nxt = nxt + actions[..., j:j + 1] * B[j]
            if kind == 1:
                nxt = np.tanh(nxt)
            r = np.zeros(observations.shape[:-1] + (1,))
            return nxt, None, r

    class NewMeanICem(RefMpcICem):
        # the override point icem.py:171,191-192: the new last row of the shifted mean from the best trajectory's last
        # predicted observation AND the row the default would have kept (tests/golden_util.py::new_mean_rule is the same rule)
        def compute_new_mean(self, obs):
            return 0.5 * self.mean[-1] + 0.25 * np.tanh(obs[:d])

    MpcICem = NewMeanICem if new_mean else RefMpcICem
    env = FakeEnv()
    ctrl = MpcICem(env=env, forward_model=FakeModel(env=env), horizon=h,
                   num_simulated_trajectories=N, factor_decrease_num=gamma,
                   cost_along_trajectory=cost_mode, verbose=False, do_visualize_plan=False,
                   action_sampler_params=dict(alpha=alpha, elites_size=K, opt_iterations=iters,
                                              init_std=init_std, use_mean_actions=use_mean,
                                              keep_previous_elites=keep, shift_elites_over_time=shift,
                                              fraction_elites_reused=xi, noise_beta=beta))
    log = {"sim_actions": [], "costs": [], "elite_idx": [], "mean": [], "std": [], "best": []}
    orig_sim = ctrl.simulate_trajectories
    orig_upd = ctrl.update_distributions

    def sim(*, obs, state, action_sequences):
        log["sim_actions"].append(np.array(action_sequences))
        return orig_sim(obs=obs, state=state, action_sequences=action_sequences)

    def upd(paths, costs):
        log["costs"].append(np.array(costs))
        log["elite_idx"].append(np.array(costs).argsort()[:ctrl.num_elites])
        log["best"].append(int(np.argmin(costs)))
        orig_upd(paths, costs)
        log["mean"].append(ctrl.mean.copy())
        log["std"].append(ctrl.std.copy())
        ec = ctrl.trajectory_cost_fn(ctrl.cost_fn, ctrl.elite_samples)
        assert np.all(np.diff(ec) >= 0), "elites not ascending"

    ctrl.simulate_trajectories = sim
    ctrl.update_distributions = upd

    colorednoise.CALLS.clear()
    orig_randn = np.random.randn
    if beta <= 0:
        # icem.py:77: white noise straight from np.random.randn(num_traj, h, d); record every draw as (z, empty, z)
        def randn(*shape):
            z = orig_randn(*shape)
            colorednoise.CALLS.append((z.copy(), np.zeros((0,)), z.copy()))
            return z
        np.random.randn = randn
    np.random.seed(seed)
    obs_rs = np.random.RandomState(1000 + seed)
    obs = 0.1 * obs_rs.randn(o)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    out = {"obs": [], "executed": [], "mean_after": []}
    for s in range(n_steps):
        out["obs"].append(obs.copy())
        a = ctrl.get_action(obs, None)
        assert a.dtype == np.float64 and a.shape == (d,)
        out["executed"].append(np.array(a))
        out["mean_after"].append(ctrl.mean.copy())
        # advance the "real" system with the same fake dynamics
        nxt, _, _ = ctrl.forward_model.predict(observations=obs[None], states=None, actions=a[None])
        obs = nxt[0]

    np.random.randn = orig_randn
    # costs must be tie-free for argsort to be well defined
    for c in log["costs"]:
        assert len(np.unique(c)) == len(c), "golden case has tied costs"

    data = dict(
        cfg=np.array([N, h, d, o, iters, seed, n_steps, kind, K], dtype=np.int64),
        cfg_f=np.array([beta, xi, gamma, alpha, init_std, bounds], dtype=np.float64),
        flags=np.array([use_mean, keep, shift], dtype=np.int64),
        env_kind=np.array(env_kind), cost_mode=np.array(cost_mode),
        A=A, B=B, low=env.action_space.low.astype(np.float64), high=env.action_space.high.astype(np.float64),
        obs=np.array(out["obs"]), executed=np.array(out["executed"]),
        n_noise_calls=np.array(len(colorednoise.CALLS)), n_iters_total=np.array(len(log["costs"])),
    )
    if new_mean:
        data["new_mean"] = np.array(1)
        data["mean_after"] = np.array(out["mean_after"])
    for i, (zr, zi, y) in enumerate(colorednoise.CALLS):
        data[f"zr_{i}"] = zr
        data[f"zi_{i}"] = zi
        if i < 2:
            data[f"y_{i}"] = y
    for i in range(len(log["costs"])):
        data[f"simact_{i}"] = log["sim_actions"][i]
        data[f"costs_{i}"] = log["costs"][i]
        data[f"elite_{i}"] = log["elite_idx"][i].astype(np.int64)
        data[f"mean_{i}"] = log["mean"][i]
        data[f"std_{i}"] = log["std"][i]
        data[f"best_{i}"] = np.array(log["best"][i])
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, "
          f"{len(colorednoise.CALLS)} noise calls, {len(log['costs'])} iterations")


def run_cem_std_case(name, *, N, h, d, o, iters, seed, n_steps, kind, like_levine, shift_means=True,
                     execute_best=True, cost_mode="sum", K=10, alpha=0.1, init_std=0.5, bounds=1.0):
    """The CEM baseline ``MpcCemStd`` (icem/controllers/mpc.py:142-327): truncated-normal sampling through
    ``scipy.stats.truncnorm.rvs``.  The uniform draws behind every sampling call are recovered by replaying the
    legacy global stream (scipy draws ONE ``uniform(size=(N, h, d))`` per call) and checked against the samples."""
    import controllers.mpc as ref_mpc
    from models.abstract_models import ForwardModelWithDefaults
    from gym import spaces
    from scipy.stats import truncnorm as sp_truncnorm
    import environments.mujoco as ref_mj

    A, B = make_model_mats(o, d)
    calls = []

    class RecordingTruncnorm:
        @staticmethod
        def rvs(a, b, loc=0, scale=1, size=1):
            state = np.random.get_state()
            out = sp_truncnorm.rvs(a, b, loc=loc, scale=scale, size=size)
            after = np.random.get_state()
            np.random.set_state(state)
            u = np.random.uniform(size=size)
            a2 = np.random.get_state()
            assert a2[2] == after[2] and np.array_equal(a2[1], after[1]), "uniform replay out of sync with scipy"
            assert np.array_equal(sp_truncnorm.ppf(u, a, b) * scale + loc, out)
            shp = tuple(size)[1:]
            calls.append((u, np.broadcast_to(np.asarray(a, dtype=np.float64), shp).copy(),
                          np.broadcast_to(np.asarray(b, dtype=np.float64), shp).copy(), out.copy()))
            return out

    class FakeEnv:
        def __init__(self):
            self.name = "FakeHalfCheetah"
            self.action_space = spaces.Box(low=-bounds * np.ones(d), high=bounds * np.ones(d))
            self.penalise_flipping = True

        def cost_fn(self, obs, act, next_obs):
            return ref_mj.HalfCheetahMaybeWithPosition.cost_fn(self, obs, act, next_obs)

        def reward_fn(self, obs, act, next_obs):
            return -self.cost_fn(obs, act, next_obs)

    class FakeModel(ForwardModelWithDefaults):
        def train(self, buffer): pass
        def save(self, path): pass
        def load(self, path): pass

        def predict(self, *, observations, states, actions):
            nxt = np.zeros_like(observations)
            for k in range(o):
                nxt = nxt + observations[..., k:k + 1] * A[k]
            for j in range(d):
                nxt = nxt + actions[..., j:j + 1] * B[j]
            if kind == 1:
                nxt = np.tanh(nxt)
            return nxt, None, np.zeros(observations.shape[:-1] + (1,))

    env = FakeEnv()
    orig_tn = ref_mpc.truncnorm
    ref_mpc.truncnorm = RecordingTruncnorm
    try:
        import io, contextlib
        ctrl = ref_mpc.MpcCemStd(env=env, forward_model=FakeModel(env=env), horizon=h, num_simulated_trajectories=N,
                                 cost_along_trajectory=cost_mode, verbose=False, do_visualize_plan=False,
                                 action_sampler_params=dict(alpha=alpha, elites_size=K, opt_iterations=iters,
                                                            init_std=init_std, shift_means=shift_means,
                                                            execute_best_elite=execute_best,
                                                            bounds_like_levine=like_levine))
        log = {"sim_actions": [], "costs": [], "elite_idx": [], "mean": [], "std": [], "lower": [], "upper": []}
        orig_sim, orig_upd = ctrl.simulate_trajectories, ctrl.update_distributions

        def sim(*, obs, state, action_sequences):
            log["sim_actions"].append(np.array(action_sequences))
            return orig_sim(obs=obs, state=state, action_sequences=action_sequences)

        def upd(paths, costs):
            log["costs"].append(np.array(costs))
            log["elite_idx"].append(np.array(costs).argsort()[:ctrl.num_elites])
            orig_upd(paths, costs)
            shp = ctrl.mean.shape
            log["mean"].append(ctrl.mean.copy())
            log["std"].append(ctrl.std.copy())
            log["lower"].append(np.broadcast_to(np.asarray(ctrl.lower, dtype=np.float64), shp).copy())
            log["upper"].append(np.broadcast_to(np.asarray(ctrl.upper, dtype=np.float64), shp).copy())

        ctrl.simulate_trajectories = sim
        ctrl.update_distributions = upd
        np.random.seed(seed)
        obs = 0.1 * np.random.RandomState(1000 + seed).randn(o)
        with contextlib.redirect_stdout(io.StringIO()):
            ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
        out = {"obs": [], "executed": [], "mean_after": [], "std_after": []}
        for _ in range(n_steps):
            out["obs"].append(obs.copy())
            a = ctrl.get_action(obs, None)
            out["executed"].append(np.array(a))
            out["mean_after"].append(ctrl.mean.copy())
            out["std_after"].append(ctrl.std.copy())
            nxt, _, _ = ctrl.forward_model.predict(observations=obs[None], states=None, actions=a[None])
            obs = nxt[0]
    finally:
        ref_mpc.truncnorm = orig_tn
    for c in log["costs"]:
        assert len(np.unique(c)) == len(c), "golden case has tied costs"
    data = dict(
        cfg=np.array([N, h, d, o, iters, seed, n_steps, kind, K], dtype=np.int64),
        cfg_f=np.array([alpha, init_std, bounds], dtype=np.float64),
        flags=np.array([like_levine, shift_means, execute_best], dtype=np.int64), cost_mode=np.array(cost_mode),
        A=A, B=B, low=env.action_space.low.astype(np.float64), high=env.action_space.high.astype(np.float64),
        obs=np.array(out["obs"]), executed=np.array(out["executed"]), mean_after=np.array(out["mean_after"]),
        std_after=np.array(out["std_after"]), n_calls=np.array(len(calls)),
    )
    for i, (u, lo, hi, x) in enumerate(calls):
        data[f"u_{i}"], data[f"lower_{i}"], data[f"upper_{i}"] = u, lo, hi
        data[f"simact_{i}"] = log["sim_actions"][i]
        assert np.array_equal(log["sim_actions"][i], x)
        data[f"costs_{i}"] = log["costs"][i]
        data[f"elite_{i}"] = log["elite_idx"][i].astype(np.int64)
        data[f"mean_{i}"], data[f"std_{i}"] = log["mean"][i], log["std"][i]
        data[f"lower_next_{i}"], data[f"upper_next_{i}"] = log["lower"][i], log["upper"][i]
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(calls)} sampling calls")


def run_random_case(name, *, N, h, d, o, freq, seed, n_steps, kind, cost_mode="sum"):
    """f-3: the random-shooting baseline MpcRandom (icem/controllers/mpc.py:86-138) with the synthetic model: records
    the unit draws behind every env.action_space.sample() call (two at construction: RndController's, unused, then
    MpcRandom's current_action), the sampled sequences, costs, argmin and executed action of every MPC step."""
    from controllers.mpc import MpcRandom
    from models.abstract_models import ForwardModelWithDefaults
    from gym import spaces
    import environments.mujoco as ref_mj
    A, B = make_model_mats(o, d)

    class Env:
        name = "fake"
        action_space = spaces.Box(-np.ones(d), np.ones(d))
        observation_space = spaces.Box(-np.inf * np.ones(o), np.inf * np.ones(o))
        penalise_flipping = True

        def cost_fn(self, obs, act, next_obs):
            return ref_mj.HalfCheetahMaybeWithPosition.cost_fn(self, obs, act, next_obs)

    class Model(ForwardModelWithDefaults):
        def train(self, buffer): pass
        def save(self, path): pass
        def load(self, path): pass

        def predict(self, *, observations, states, actions):
            nxt = np.zeros_like(observations)
            for k in range(o):
                nxt = nxt + observations[..., k:k + 1] * A[k]
            for j in range(d):
                nxt = nxt + actions[..., j:j + 1] * B[j]
            if kind == 1:
                nxt = np.tanh(nxt)
            return nxt, None, np.zeros(observations.shape[:-1] + (1,))

    del spaces.SAMPLE_LOG[:]
    np.random.seed(seed)
    env = Env()
    params = types.SimpleNamespace(action_change_frequency=freq)

    class Runnable(MpcRandom):   # the reference class leaves these two abstract (it cannot be instantiated as shipped)
        def beginning_of_rollout(self, *, observation, state=None, mode): pass
        def end_of_rollout(self, total_time, total_return, mode): pass

    ctrl = Runnable(env=env, forward_model=Model(env=env), horizon=h, num_simulated_trajectories=N,
                    cost_along_trajectory=cost_mode, action_sampler_params=params)
    n_init = len(spaces.SAMPLE_LOG)
    data = dict(N=N, h=h, d=d, o=o, freq=freq, kind=kind, A=A, B=B, n_steps=n_steps, cost_mode=cost_mode, n_init_draws=n_init)
    orig = ctrl.trajectory_cost_fn
    log = {}

    def spy(cost_fn, paths):
        c = orig(cost_fn, paths)
        log["costs"], log["actions"] = np.array(c), np.array(paths.as_array("actions"))
        return c
    ctrl.trajectory_cost_fn = spy
    rs = np.random.RandomState(seed + 1)
    for sidx in range(n_steps):
        obs = 0.1 * rs.randn(o)
        a = ctrl.get_action(obs, None)
        data[f"obs_{sidx}"], data[f"executed_{sidx}"] = obs, np.array(a)
        data[f"actions_{sidx}"], data[f"costs_{sidx}"] = log["actions"], log["costs"]
        data[f"best_{sidx}"] = int(np.argmin(log["costs"]))
    data["u"] = np.array(spaces.SAMPLE_LOG)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(spaces.SAMPLE_LOG)} draws ({n_init} at construction)")


def cost_fn_vectors():
    """Direct input/output vectors of the reference's two cost functions."""
    import environments.mujoco as ref_mj
    rs = np.random.RandomState(7)
    fake = types.SimpleNamespace(penalise_flipping=True)
    fake_nf = types.SimpleNamespace(penalise_flipping=False)
    o17 = rs.randn(16, 10, 17) * 1.5
    o18 = rs.randn(16, 10, 18) * 1.5
    a6 = rs.uniform(-1, 1, (16, 10, 6))
    o378 = rs.randn(4, 10, 378)
    a17 = rs.uniform(-0.4, 0.4, (4, 10, 17))
    HC = ref_mj.HalfCheetahMaybeWithPosition
    data = dict(
        o17=o17, o18=o18, a6=a6, o378=o378, a17=a17,
        hc17=HC.cost_fn(fake, o17, a6, None), hc18=HC.cost_fn(fake, o18, a6, None),
        hc17_noflip=HC.cost_fn(fake_nf, o17, a6, None),
        hc17_single=HC.cost_fn(fake, o17[0, 0], a6[0, 0], None),
        hs=ref_mj.HumanoidStandup.cost_fn(fake, o378, a17, None),
    )
    path = os.path.join(OUT, "cost_fn_vectors.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def env_cost_vectors():
    """f-4: input/output vectors of the reference's remaining parametric cost functions (Ant, Hopper, Humanoid,
    Reacher in environments/mujoco.py; FetchPickAndPlace, FetchReach in environments/robotics.py), each called
    unbound on a stand-in ``self`` carrying the attributes it reads (the gym v3 defaults)."""
    import environments.mujoco as ref_mj
    import environments.robotics as ref_rb
    from environments.abstract_environments import MaskedGoalSpaceEnvironmentInterface as MG
    rs = np.random.RandomState(9)
    n, h = 8, 6

    def bind(ns, cls, *names):
        for nm in names:
            setattr(ns, nm, types.MethodType(getattr(cls, nm), ns))
        return ns

    data = {}
    # Ant: o = 113 (positions included), z = obs[2] around the healthy range [0.2, 1.0], a few non-finite rows
    ant = bind(types.SimpleNamespace(_healthy_z_range=(0.2, 1.0), dt=0.05, _ctrl_cost_weight=0.5), ref_mj.Ant,
               "are_states_unhealthy")
    o = rs.randn(n, h, 113)
    o[..., 2] = rs.uniform(0.0, 1.3, (n, h))
    o[0, 1, 2], o[0, 2, 2] = 0.2, 1.0            # the closed ends of the range
    o[1, 0, 50], o[1, 3, 7] = np.inf, np.nan
    nx = o + 0.05 * rs.randn(n, h, 113)
    a = rs.uniform(-1, 1, (n, h, 8))
    data.update(ant_obs=o, ant_next=nx, ant_act=a, ant=ref_mj.Ant.cost_fn(ant, o, a, nx))
    # Hopper: o = 12, z = obs[1] around 0.7, state box (-100, 100) over obs[2:]
    hop = bind(types.SimpleNamespace(_healthy_z_range=(0.7, float("inf")), _healthy_state_range=(-100.0, 100.0),
                                     _healthy_angle_range=(-0.2, 0.2), dt=0.008, _ctrl_cost_weight=1e-3),
               ref_mj.Hopper, "unhealthy_states")
    o = rs.randn(n, h, 12)
    o[..., 1] = rs.uniform(0.4, 1.4, (n, h))
    o[0, 1, 1] = 0.7                              # open end
    o[2, 0, 5], o[2, 1, 11], o[2, 2, 0] = 150.0, -100.0, 500.0   # outside / on the edge of the box; obs[0] is not in it
    o[3, 0, 4] = np.nan
    nx = o + 0.01 * rs.randn(n, h, 12)
    a = rs.uniform(-1, 1, (n, h, 3))
    data.update(hopper_obs=o, hopper_next=nx, hopper_act=a, hopper=ref_mj.Hopper.cost_fn(hop, o, a, nx))
    # Humanoid: positions excluded (o = 376 here is irrelevant to the cost: any width), z = obs[0], velocity obs[nq-2]
    for excl, o_dim, tag in ((True, 40, "humanoid_excl"), (False, 42, "humanoid_incl")):
        hum = bind(types.SimpleNamespace(_healthy_z_range=(1.0, 2.0), _exclude_current_positions_from_observation=excl,
                                         model=types.SimpleNamespace(nq=24), _forward_reward_weight=1.25,
                                         _ctrl_cost_weight=0.1), ref_mj.Humanoid, "unhealthy_states")
        o = rs.randn(n, h, o_dim)
        o[..., 0 if excl else 2] = rs.uniform(0.6, 2.4, (n, h))
        o[4, 4, 9] = -np.inf
        a = rs.uniform(-0.4, 0.4, (n, h, 17))
        data.update({tag + "_obs": o, tag + "_act": a, tag: ref_mj.Humanoid.cost_fn(hum, o, a, o)})
    # Reacher: norm of the last three observation entries
    o = rs.randn(n, h, 11)
    a = rs.uniform(-1, 1, (n, h, 2))
    data.update(reacher_obs=o, reacher_act=a, reacher=ref_mj.Reacher.cost_fn(None, o, a, None))
    # FetchPickAndPlace (obs 25 + goal 3) and FetchReach (obs 10 + goal 3), dense / sparse / shaped
    for tag, cls, olen, ach in (("fpp", ref_rb.FetchPickAndPlace, 25, [3, 4, 5]), ("freach", ref_rb.FetchReach, 10, [0, 1, 2])):
        o = 0.1 * rs.randn(n * h, olen + 3)      # robotics cost_fn indexes observation[:, :3] -> 2-D input
        near = np.arange(0, n * h, 2)            # every other row: goal (and gripper) within ~threshold of the object
        o[near, olen:] = o[near][:, ach] + 0.03 * rs.randn(len(near), 3)
        if tag == "fpp":
            o[near[::2], 0:3] = o[near[::2], 3:6] + 0.03 * rs.randn(len(near[::2]), 3)
        a = rs.uniform(-1, 1, (n * h, 4))
        data.update({tag + "_obs": o, tag + "_act": a})
        for sparse in (False, True):
            for shaped in ((False, True) if tag == "fpp" else (False,)):
                ns = bind(types.SimpleNamespace(goal_idx=np.arange(olen, olen + 3), achieved_goal_idx=ach, sparse=sparse,
                                                threshold=0.05, shaped_reward=shaped), MG,
                          "goal_from_observation", "achieved_goal_from_observation")
                key = f"{tag}_{'sparse' if sparse else 'dense'}{'_shaped' if shaped else ''}"
                data[key] = np.asarray(cls.cost_fn(ns, o, a, None))
    # Door / Relocate (hand manipulation suite): environments/mjenvs.py imports the un-vendored mj_envs submodule at
    # module level -> stand-in modules for those four imports, then the cost functions are called unbound, one
    # trajectory at a time as trajectory_cost_fn does (Door flattens door_pos: 2-D input only)
    pk = "environments.mj_envs"
    for name, attrs in ((pk, {}), (pk + ".mj_envs", {}), (pk + ".mj_envs.hand_manipulation_suite", {}),
                        (pk + ".mj_envs.hand_manipulation_suite.door_v0", dict(DoorEnvV0=object, ADD_BONUS_REWARDS=True)),
                        (pk + ".mj_envs.hand_manipulation_suite.relocate_v0", dict(RelocateEnvV0=object, ADD_BONUS_REWARDS=True)),
                        (pk + ".mj_envs.hand_manipulation_suite.hammer_v0", dict(HammerEnvV0=object, ADD_BONUS_REWARDS=True))):
        mod = types.ModuleType(name)
        mod.__path__ = []
        mod.__dict__.update(attrs)
        sys.modules[name] = mod
    import environments.mjenvs as ref_mjenvs
    o_dim = 39
    for tag, shaped, bonus in (("door", True, True), ("door_plain", False, False)):
        door = types.SimpleNamespace(door_pos_idx=np.arange(28, 29), palm_pos_idx=np.arange(29, 32),
                                     handle_pos_idx=np.arange(32, 35), qv_start_idx=30, shaped_reward=shaped,
                                     add_bonus_rewards=bonus)
        o = 0.3 * rs.randn(n, h, o_dim)
        o[..., 28] = rs.uniform(-0.2, 1.7, (n, h))    # hinge angle across the 0.2 / 1.0 / 1.35 bonus steps
        a = rs.uniform(-1, 1, (n, h, 28))
        data.update({tag + "_obs": o, tag + "_act": a,
                     tag: np.stack([ref_mjenvs.Door.cost_fn(door, o[i], a[i], None) for i in range(n)])})
    for tag, bonus in (("relocate", True), ("relocate_plain", False)):
        rel = types.SimpleNamespace(palm_pos_minus_obj_pos_idx=np.arange(30, 33), palm_pos_minus_target_pos_idx=np.arange(33, 36),
                                    obj_pos_minus_target_pos_idx=np.arange(36, 39), add_bonus_rewards=bonus)
        o = 0.2 * rs.randn(n, h, o_dim)
        o[::2, :, 36:39] = 0.04 * rs.randn(n // 2, h, 3)   # object near the target (both closeness bonuses) ...
        o[1::3, :, 38] = rs.uniform(0.03, 0.08, (len(range(1, n, 3)), h))   # ... and just lifted / not lifted
        a = rs.uniform(-1, 1, (n, h, 30))
        data.update({tag + "_obs": o, tag + "_act": a,
                     tag: np.stack([ref_mjenvs.Relocate.cost_fn(rel, o[i], a[i], None) for i in range(n)])})
    path = os.path.join(OUT, "env_cost_vectors.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def open_loop_policy_vectors():
    """The order in which the reference's ``OpenLoopPolicy`` (abstract_controller.py:153-184 over
    ``ArrayIteratorParallelRowwise``, controllers/utils.py:18-51) hands out a ``[p,h,d]`` array, for every way a forward
    model asks: 1-D observations (``GroundTruthModel.predict_n_steps``, gt_model.py:84-90), ``k < p`` rows, all rows, the
    sub-policies of ``get_parallel_policy_copy`` (gt_par_model.py:77-80) -- each walked until the reference raises; the
    exception's type name is recorded too."""
    from controllers.abstract_controller import OpenLoopPolicy
    data, cases = {}, []
    for (p, h, d, k) in [(5, 4, 2, 1), (6, 3, 2, 2), (6, 3, 2, 6), (4, 3, 1, 3), (1, 3, 2, 1), (3, 1, 2, 2), (7, 5, 3, 7),
                         (2, 3, 2, 3)]:
        seq = np.arange(p * h * d, dtype=np.float64).reshape(p, h, d)
        obs = np.zeros(4) if k == 1 else np.zeros((k, 4))
        outs, err = [], "none"
        try:
            pol = OpenLoopPolicy(seq)
            for _ in range(p * h + 2):
                outs.append(np.array(pol.get_action(obs, None), dtype=np.float64))
        except Exception as e:  # noqa: BLE001 -- the type IS the datum
            err = type(e).__name__
        name = f"olp_p{p}_h{h}_d{d}_k{k}"
        cases.append(name)
        data[name + "_cfg"] = np.array([p, h, d, k], dtype=np.int64)
        data[name + "_ndim"] = np.array([o.ndim for o in outs], dtype=np.int64)
        data[name + "_rows"] = np.array([1 if o.ndim == 1 else o.shape[0] for o in outs], dtype=np.int64)
        data[name + "_flat"] = np.concatenate([o.reshape(-1) for o in outs]) if outs else np.zeros(0)
        data[name + "_err"] = np.array(err)
    # sub-policies as ParallelGroundTruthModel builds them: array_split chunks, each walked one row at a time
    p, h, d, workers = 11, 4, 2, 4
    seq = np.arange(p * h * d, dtype=np.float64).reshape(p, h, d)
    pol = OpenLoopPolicy(seq)
    chunks = [c for c in np.array_split(range(p), workers) if len(c) > 0]
    flat = []
    for c in chunks:
        sub = pol.get_parallel_policy_copy(c)
        for _ in range(len(c) * h):
            flat.append(np.array(sub.get_action(np.zeros(4), None)))
    data["olp_chunks_cfg"] = np.array([p, h, d, workers], dtype=np.int64)
    data["olp_chunks_flat"] = np.concatenate(flat)
    data["cases"] = np.array(cases)
    path = os.path.join(OUT, "open_loop_policy_vectors.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    install_stubs()
    if "--new-mean-only" in sys.argv:   # (round 6: the one fixture added without rewriting the others)
        run_case("newmean_n48", N=48, h=10, d=4, o=17, beta=1.0, iters=3, seed=17,
                 n_steps=3, kind=1, env_kind="halfcheetah", new_mean=True)
        return
    cost_fn_vectors()
    env_cost_vectors()
    open_loop_policy_vectors()
    # C1-shaped: HalfCheetah shapes, beta=0.25, 3 iterations (BASELINE.json configs[0])
    run_case("c1_halfcheetah_n128", N=128, h=30, d=6, o=17, beta=0.25, iters=3, seed=11,
             n_steps=3, kind=0, env_kind="halfcheetah")
    # HumanoidStandup-shaped actions (d=17, bounds 0.4, beta=2.0); o kept small (synthetic latent)
    run_case("humanoid_n40_d17", N=40, h=30, d=17, o=24, beta=2.0, iters=3, seed=12,
             n_steps=2, kind=1, env_kind="humanoid", bounds=0.4)
    # PlaNet-horizon shapes h=12
    run_case("h12_n64", N=64, h=12, d=6, o=17, beta=0.25, iters=3, seed=13,
             n_steps=3, kind=1, env_kind="halfcheetah")
    # odd horizon: no Nyquist bin
    run_case("h13_odd_n48", N=48, h=13, d=4, o=17, beta=1.0, iters=4, seed=14,
             n_steps=3, kind=0, env_kind="halfcheetah", cost_mode="best")
    # 'final' cost mode, no elite keeping / shifting, no mean action
    run_case("final_noreuse_n40", N=40, h=10, d=3, o=17, beta=3.0, iters=3, seed=15,
             n_steps=2, kind=1, env_kind="halfcheetah", cost_mode="final",
             use_mean=False, keep=False, shift=False)
    # beta = 0: the white-noise branch (np.random.randn(N, h, d), icem.py:77), elites shifted and kept
    run_case("white_beta0_n64", N=64, h=30, d=6, o=17, beta=0.0, iters=3, seed=16,
             n_steps=3, kind=0, env_kind="halfcheetah")
    # a subclass overriding compute_new_mean (icem.py:171,191-192): the last mean row from the last predicted observation
    run_case("newmean_n48", N=48, h=10, d=4, o=17, beta=1.0, iters=3, seed=17,
             n_steps=3, kind=1, env_kind="halfcheetah", new_mean=True)
    # the CEM baseline MpcCemStd (truncated normal): bounds from the action space, and "like Levine" (+-2 sigma, std capped)
    run_cem_std_case("cemstd_bounds_n48", N=48, h=12, d=6, o=17, iters=3, seed=21, n_steps=3, kind=0, like_levine=False)
    run_random_case("random_n24", N=24, h=10, d=4, o=17, freq=3, seed=31, n_steps=3, kind=1)
    run_cem_std_case("cemstd_levine_n40", N=40, h=10, d=4, o=17, iters=4, seed=22, n_steps=2, kind=1, like_levine=True,
                     execute_best=False, cost_mode="best")


if __name__ == "__main__":
    main()
