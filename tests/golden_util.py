"""Helpers to load the golden fixtures written by tests/golden/make_golden.py."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not p.endswith(("cost_vectors.npz", "cost_fn_vectors.npz", "open_loop_policy_vectors.npz")))
CASES = [n for n in _ALL if not n.startswith(("cemstd_", "random_", "newmean_"))]        # MpcICem runs
NEWMEAN_CASES = [n for n in _ALL if n.startswith("newmean_")]   # MpcICem subclasses that override compute_new_mean (icem.py:171,191-192)


def new_mean_rule(last_predicted_obs, kept_row):
    """The override the ``newmean_*`` fixtures were recorded with (tests/golden/make_golden.py::NewMeanICem)."""
    d = kept_row.shape[0]
    return 0.5 * kept_row + 0.25 * np.tanh(np.asarray(last_predicted_obs)[:d])

RANDOM_CASES = [n for n in _ALL if n.startswith("random_")]     # MpcRandom runs (random shooting baseline)
CEMSTD_CASES = [n for n in _ALL if n.startswith("cemstd_")]     # MpcCemStd runs (truncated-normal CEM baseline)


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        (self.N, self.h, self.d, self.o, self.iters, self.seed, self.n_steps,
         self.kind, self.K) = [int(v) for v in z["cfg"]]
        (self.beta, self.xi, self.gamma, self.alpha, self.init_std, self.bounds) = [float(v) for v in z["cfg_f"]]
        self.use_mean, self.keep, self.shift = [bool(v) for v in z["flags"]]
        self.env_kind = str(z["env_kind"])
        self.cost_mode = str(z["cost_mode"])
        self.A, self.B = z["A"], z["B"]
        self.low, self.high = z["low"], z["high"]
        self.obs, self.executed = z["obs"], z["executed"]
        self.n_noise_calls = int(z["n_noise_calls"])
        self.n_iters_total = int(z["n_iters_total"])
        self.new_mean = "new_mean" in z.files
        self.mean_after = z["mean_after"] if self.new_mean else None   # [n_steps, h, d]: ctrl.mean behind every get_action

    def noise(self, i):
        return self.z[f"zr_{i}"], self.z[f"zi_{i}"]

    def it(self, i):
        z = self.z
        return dict(simact=z[f"simact_{i}"], costs=z[f"costs_{i}"], elite=z[f"elite_{i}"],
                    mean=z[f"mean_{i}"], std=z[f"std_{i}"], best=int(z[f"best_{i}"]))


class GoldenCemStd:
    """A recorded run of the reference's ``MpcCemStd`` (tests/golden/make_golden.py::run_cem_std_case)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        (self.N, self.h, self.d, self.o, self.iters, self.seed, self.n_steps, self.kind, self.K) = [int(v) for v in z["cfg"]]
        self.alpha, self.init_std, self.bounds = [float(v) for v in z["cfg_f"]]
        self.like_levine, self.shift_means, self.execute_best = [bool(v) for v in z["flags"]]
        self.cost_mode = str(z["cost_mode"])
        self.A, self.B, self.low, self.high = z["A"], z["B"], z["low"], z["high"]
        self.obs, self.executed = z["obs"], z["executed"]
        self.mean_after, self.std_after = z["mean_after"], z["std_after"]
        self.n_calls = int(z["n_calls"])

    def call(self, i):
        z = self.z
        return {k: z[f"{k}_{i}"] for k in ("u", "lower", "upper", "simact", "costs", "elite", "mean", "std", "lower_next", "upper_next")}


class GoldenRandom:
    """A recorded run of the reference's MpcRandom (tests/golden/make_golden.py::run_random_case)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        for k in ("N", "h", "d", "o", "freq", "kind", "n_steps", "n_init_draws"):
            setattr(self, k, int(z[k]))
        self.cost_mode = str(z["cost_mode"])
        self.A, self.B, self.u = z["A"], z["B"], z["u"]
        self.low, self.high = -np.ones(self.d), np.ones(self.d)

    def block_uniforms(self, first_block, n_blocks):
        """Draws of blocks first_block.. : block 0 is the LAST construction-time draw (MpcRandom.current_action; the
        one before it is RndController.previous_action, never used)."""
        lo = self.n_init_draws - 1 + first_block
        return self.u[lo:lo + n_blocks]

    def step(self, s):
        return {k: self.z[f"{k}_{s}"] for k in ("obs", "executed", "actions", "costs", "best")}
