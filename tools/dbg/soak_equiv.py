"""Randomised soak: icem_plan_step (merges folded into the next launch, ping-pong buffers) against the split API
(one merge launch per iteration) must agree bit for bit -- over random populations, elite counts, iteration counts,
flags, seeds, shapes and cost modes.  usage: soak_equiv.py [n_cases] [seed] [large]
("large": populations of 8 200 .. 140 000 rows with slow decay, so that most cases take the noise-ahead launches of
k_rollout_ahead.hip on the icem_plan_step side; the split API stays on the sampler + rollout pair)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner  # noqa: E402

SHAPES = [(30, 6, 17), (30, 6, 18), (12, 6, 17), (13, 4, 17), (30, 17, 24)]


def np_(t):
    return t.detach().cpu().numpy()


LARGE = False


def one(rs, case):
    h, d, o = SHAPES[rs.randint(len(SHAPES))]
    N = int(rs.choice([rs.randint(2, 200), rs.randint(200, 5000), rs.randint(5000, 40000), rs.randint(40000, 70000)]))
    if LARGE:
        N = int(rs.choice([rs.randint(8200, 20000), rs.randint(20000, 70000), rs.randint(70000, 140000)]))
    if d == 17:
        N = min(N, 20000 if not LARGE else 40000)
    K = int(rs.randint(2, 12))
    iters = int(rs.randint(1, 6)) if not LARGE else int(rs.randint(2, 6))
    kind = int(rs.randint(2))
    flags = dict(use_mean_actions=bool(rs.randint(2)), keep_previous_elites=bool(rs.randint(2)), shift_elites=bool(rs.randint(2)))
    mode = ["sum", "best", "final"][rs.randint(3)]
    beta = float(rs.choice([0.0, 0.25, 1.0, 2.5]))
    seed = int(rs.randint(1 << 30))
    cfgkw = dict(horizon=h, act_dim=d, num_traj=N, elites_size=K, opt_iters=iters, dtype="f32", seed=seed, cost_mode=mode,
                 noise_beta=beta, factor_decrease=float(rs.choice([1.0, 1.25, 2.0] if not LARGE else [1.0, 1.1, 1.25])), **flags)
    low, high = -np.ones(d), np.ones(d)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    pls = []
    for _ in range(2):
        pl = IcemPlanner(IcemConfig(**cfgkw), low, high)
        pl.set_model(kind, model.A, model.B)
        pl.set_cost(0.1, int(rs.randint(o)) if False else 3, -1.0, 1, 10.0, 0.5)
        pl.reset()
        pls.append(pl)
    for step in range(3):
        obs = 0.2 * rs.randn(o)
        a0 = np_(pls[0].plan_step(obs))
        a1 = np_(pls[1].plan_step(obs, on_iteration=lambda it: None))
        ok = np.array_equal(a0, a1)
        for name in ("mean", "std", "best_cost"):
            ok &= np.array_equal(np_(getattr(pls[0], name)), np_(getattr(pls[1], name)))
        (ea0, ec0), (ea1, ec1) = pls[0].current_elites(), pls[1].current_elites()
        ok &= np.array_equal(np_(ea0), np_(ea1)) and np.array_equal(np_(ec0), np_(ec1))
        n_last = pls[0].population_sizes[-1]
        ok &= np.array_equal(np_(pls[0].actions[:n_last]), np_(pls[1].actions[:n_last]))
        ok &= np.array_equal(np_(pls[0].costs[:n_last]), np_(pls[1].costs[:n_last]))
        if not ok:
            print("MISMATCH case", case, "step", step, cfgkw, "kind", kind, flush=True)
            return False
    return True


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = sum(not one(rs, c) for c in range(n))
    torch.cuda.synchronize()
    print(f"soak: {n} random configurations, {bad} mismatching")
    sys.exit(1 if bad else 0)
