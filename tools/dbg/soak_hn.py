"""Randomised soak of TileHN's wave arrangements (rollout_hn_split_kernel / _pair_ / one wave per tile: ICEM_HN_SPLIT=0,
ICEM_HN_PAIR=0) on the Door / Relocate / FetchPickAndPlace shapes: random populations (tile counts around every boundary:
256, 272, 512), elite counts, iteration counts, cost modes, models -- every buffer bit for bit.
usage (GPU box): python tools/dbg/soak_hn.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402
from icem_amd import envs as E  # noqa: E402
from icem_amd import _lib as _LENV  # noqa: E402
_LENV.follow_environment()   # this tool flips ICEM_<NAME> variables: mapped onto icem_set_option per planner (the library reads no environment)

ENVS = {"door": E.door_env, "relocate": E.relocate_env, "fpp": E.fetch_pick_and_place_env}


def run(env, model, cfg_kw, obs_seq, variant):
    for k in ("ICEM_HN_SPLIT", "ICEM_HN_PAIR"):
        os.environ.pop(k, None)
    os.environ.update(variant)
    pl = IcemPlanner(IcemConfig(**cfg_kw), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    assert pl.tile_arith == 1
    out = []
    for obs in obs_seq:
        a = pl.plan_step(obs).cpu().numpy().copy()
        ea, ec = pl.current_elites()
        out.append((a, pl.costs.cpu().numpy().copy(), ea.cpu().numpy().copy(), ec.cpu().numpy().copy(),
                    pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy()))
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        name = str(rs.choice(list(ENVS)))
        env = ENVS[name]()
        o, d = env.obs_dim, env.action_space.shape[0]
        N = int(rs.choice([rs.randint(2, 300), rs.randint(3900, 4400), rs.randint(4000, 4200), rs.randint(8000, 8400), rs.randint(300, 12000)]))
        cfg = dict(horizon=30, act_dim=d, num_traj=N, elites_size=int(rs.choice([2, 5, 10, 16])), opt_iters=int(rs.randint(1, 4)), dtype="f32",
                   seed=int(rs.randint(1, 1 << 30)), cost_mode=str(rs.choice(["sum", "best", "final"])), noise_beta=float(rs.choice([0.25, 2.5, 3.5])),
                   shift_elites=bool(rs.randint(0, 2)), keep_previous_elites=bool(rs.randint(0, 2)), factor_decrease=float(rs.choice([1.0, 1.25, 2.0])))
        model = DeviceSyntheticModel.make(o, d, kind=int(rs.randint(0, 2)))
        obs_seq = [0.2 * rs.randn(o) for _ in range(3)]
        res = [run(env, model, cfg, obs_seq, v) for v in ({}, {"ICEM_HN_SPLIT": "0"}, {"ICEM_HN_PAIR": "0"})]
        same = all(np.array_equal(u, v, equal_nan=True) for other in res[:2] for x, y in zip(other, res[2]) for u, v in zip(x, y))
        if not same:
            bad += 1
            print("MISMATCH case", case, name, {k: cfg[k] for k in ("num_traj", "elites_size", "opt_iters", "cost_mode", "shift_elites", "factor_decrease")})
    print(f"TileHN soak: {n_cases} random configurations x 3 wave arrangements, {bad} mismatching")


if __name__ == "__main__":
    main()
