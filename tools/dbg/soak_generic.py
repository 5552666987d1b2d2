"""Randomised soak of the strict-parity / generic path: the row-of-lanes rollout + one-launch selection (and the four-lane
sampler's counterpart with ICEM_GK_SAMPLE=thread on both sides) against the round-4 forms (ICEM_GK_ROLLOUT=thread,
ICEM_GK_SELECT=0) must agree bit for bit over random shapes, populations, elite counts, flags and cost modes, f64 and f32.
usage (GPU box): python tools/dbg/soak_generic.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)


def run(cfg_kw, model, cost, obs_seq, old):
    for k in ("ICEM_GK_ROLLOUT", "ICEM_GK_SELECT"):
        os.environ.pop(k, None)
    if old:
        os.environ["ICEM_GK_ROLLOUT"] = "thread"
        os.environ["ICEM_GK_SELECT"] = "0"
    d = cfg_kw["act_dim"]
    pl = IcemPlanner(IcemConfig(**cfg_kw), -np.ones(d), np.ones(d))
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost(*cost)
    pl.reset()
    out = []
    for obs in obs_seq:
        a = pl.plan_step(obs).cpu().numpy().copy()
        ea, ec = pl.current_elites()
        out.append((a, pl.costs.cpu().numpy().copy(), ea.cpu().numpy().copy(), ec.cpu().numpy().copy(),
                    pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy()))
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    os.environ["ICEM_GK_SAMPLE"] = "thread"   # the same sampler on both sides (the four-lane one sums in another order)
    bad = 0
    for case in range(n_cases):
        dtype = "f64" if rs.rand() < 0.7 else "f32"
        if dtype == "f32":
            os.environ["ICEM_DISABLE_FAST"] = "1"
        else:
            os.environ.pop("ICEM_DISABLE_FAST", None)
        o = int(rs.choice([3, 8, 9, 16, 17, 18, 20, 24, 29, 32]))
        d = int(rs.choice([1, 2, 3, 6, 8, 9, 12, 17]))
        h = int(rs.choice([2, 5, 12, 13, 30, 33, 40]))
        N = int(rs.choice([rs.randint(2, 40), rs.randint(40, 600), rs.randint(600, 5000)]))
        K = int(rs.choice([1, 2, 3, 10, 16, 17, 31]))
        iters = int(rs.randint(1, 4))
        kind = int(rs.randint(0, 2))
        cfg = dict(horizon=h, act_dim=d, num_traj=N, elites_size=K, opt_iters=iters, dtype=dtype, seed=int(rs.randint(1, 1 << 30)),
                   cost_mode=str(rs.choice(["sum", "best", "final"])), noise_beta=float(rs.choice([0.0, 0.25, 2.0])),
                   use_mean_actions=bool(rs.randint(0, 2)), keep_previous_elites=bool(rs.randint(0, 2)), shift_elites=bool(rs.randint(0, 2)),
                   fraction_reused=float(rs.choice([0.3, 0.5, 1.0])))
        model = DeviceSyntheticModel.make(o, d, kind=kind)
        cost = (float(rs.choice([0.0, 0.1])), int(rs.randint(0, o)), float(rs.choice([0.0, -1.0])), int(rs.randint(-1, o)), 10.0, 0.3)
        obs_seq = [0.3 * rs.randn(o) for _ in range(2)]
        try:
            new, old = run(cfg, model, cost, obs_seq, False), run(cfg, model, cost, obs_seq, True)
        except Exception as ex:   # a configuration the library refuses: the same on both sides
            print("case", case, "skipped:", str(ex)[:100])
            continue
        same = all(np.array_equal(u, v, equal_nan=True) for x, y in zip(new, old) for u, v in zip(x, y))
        if not same:
            bad += 1
            print("MISMATCH case", case, dtype, dict(o=o, d=d, h=h, N=N, K=K, iters=iters, kind=kind), cfg["cost_mode"], cost)
    print(f"generic soak: {n_cases} random configurations, {bad} mismatching")


if __name__ == "__main__":
    main()
