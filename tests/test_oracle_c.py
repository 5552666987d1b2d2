"""The plain-C oracle (cpu_baseline 'port') against the NumPy oracle, which is pinned to the
reference's golden vectors."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import icem_oracle as O


@pytest.fixture(scope="module")
def lib():
    return CO.load()


@pytest.mark.parametrize("h,beta", [(30, 0.25), (13, 1.0), (12, 2.0)])
def test_c_noise_tables(lib, h, beta):
    F = h // 2 + 1
    cr, ci = np.zeros((F, h)), np.zeros((F, h))
    lib.icem_c_noise_tables(h, beta, cr, ci)
    Cr, Ci = O.synthesis_matrices(h, beta)
    np.testing.assert_allclose(cr, Cr, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(ci, Ci, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("h,d,rounds", [(30, 6, 10), (13, 4, 7), (12, 17, 10)])
def test_c_sample_matches_numpy_oracle(lib, h, d, rounds):
    n, seed, off, first, beta = 50, 0x123456789ABCDEF, (3 << 32) | 9, 77, 0.5
    rs = np.random.RandomState(0)
    mean, std = rs.uniform(-0.2, 0.2, (h, d)), rs.uniform(0.2, 0.6, (h, d))
    low, high = -np.ones(d), np.ones(d)
    out = np.zeros((n, h, d))
    lib.icem_c_sample_clip(n, h, d, beta, seed, off, first, rounds, mean, std, low, high, 0, out)
    zr, zi = O.philox_white_noise(seed, off, n, d, h, first_index=first, rounds=rounds)
    ref = O.sample_action_sequences(mean, std, low, high, beta, zr, zi)
    np.testing.assert_allclose(out, ref, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("kind,mode", [(0, 0), (1, 1), (1, 2)])
def test_c_rollout_topk_refit(lib, kind, mode):
    n, h, d, o, K = 400, 30, 6, 17, 10
    rs = np.random.RandomState(1)
    m = O.SyntheticModel.make(o, d, kind)
    cs = O.CostSpec.halfcheetah(o)
    act = rs.uniform(-1, 1, (n, h, d))
    obs0 = 0.5 * rs.randn(o)
    costs = np.zeros(n)
    lib.icem_c_rollout_cost(n, h, d, o, kind, mode, CO.c64(m.A), CO.c64(m.B), obs0, act, cs.ctrl_weight, cs.lin_idx,
                            cs.lin_weight, cs.flip_idx, cs.flip_penalty, cs.flip_thresh, costs)
    ref = O.rollout_costs(m, cs, obs0, act, mode=["sum", "best", "final"][mode])
    np.testing.assert_allclose(costs, ref, rtol=1e-11, atol=1e-12)
    costs[5] = costs[9]  # a tie: lower index first
    costs[17] = np.nan
    idx, out = np.zeros(K, dtype=np.int32), np.zeros(K)
    lib.icem_c_topk(n, K, costs, idx, out)
    assert np.array_equal(idx, O.topk_sorted(costs, K))
    mean, std = rs.randn(h, d) * 0.1, rs.uniform(0.1, 0.5, (h, d))
    rm, rstd = O.refit(act[idx], mean, std, 0.1)
    lib.icem_c_refit(K, h * d, 0.1, act, idx, mean, std)
    np.testing.assert_allclose(mean, rm, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(std, rstd, rtol=1e-12, atol=1e-14)


def test_c_threads(lib):
    assert lib.icem_c_num_threads() >= 1
