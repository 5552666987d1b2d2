"""Pin the CPU oracle against outputs of the reference itself (tests/golden)."""
import numpy as np
import pytest

from oracle import icem_oracle as O
from golden_util import CASES, NEWMEAN_CASES, Golden, GOLDEN, new_mean_rule
import os


def _oracle_for(g: Golden, noise_fn):
    p = O.IcemParams(horizon=g.h, num_simulated_trajectories=g.N, factor_decrease_num=g.gamma,
                     cost_along_trajectory=g.cost_mode, alpha=g.alpha, elites_size=g.K,
                     opt_iterations=g.iters, init_std=g.init_std, use_mean_actions=g.use_mean,
                     keep_previous_elites=g.keep, shift_elites_over_time=g.shift,
                     fraction_elites_reused=g.xi, noise_beta=g.beta)
    model = O.SyntheticModel(g.A, g.B, g.kind)
    cost = O.CostSpec.halfcheetah(g.o) if g.env_kind == "halfcheetah" else O.CostSpec.humanoid_standup()

    def rollout_cost(obs, actions):
        observations = O.rollout_observations(model, obs, actions)
        return O.trajectory_costs(cost, observations, actions, g.cost_mode)

    if g.new_mean:   # the run of a subclass that overrides compute_new_mean (icem.py:171,191-192)
        return O.IcemOracle(p, g.low, g.high, rollout_cost, noise_fn, compute_new_mean=new_mean_rule,
                            rollout_obs=lambda obs, actions: O.rollout_observations(model, obs, actions))
    return O.IcemOracle(p, g.low, g.high, rollout_cost, noise_fn)


def test_cases_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("name", CASES)
def test_colored_noise_matches_reference_call(name):
    """colored_from_white(z) == what the reference's sampling call returned."""
    g = Golden(name)
    if g.beta <= 0:  # white branch (icem.py:77): the recorded draw IS the sample, nothing to synthesise
        for i in range(2):
            zr, zi = g.noise(i)
            assert zr.shape[1:] == (g.h, g.d) and zi.size == 0 and np.array_equal(zr, g.z[f"y_{i}"])
        return
    for i in range(2):
        zr, zi = g.noise(i)
        y = O.colored_from_white(g.beta, g.h, zr, zi)
        assert np.array_equal(y, g.z[f"y_{i}"])
        # synthesis-matrix formulation (the one the HIP kernel uses) agrees to rounding
        Cr, Ci = O.synthesis_matrices(g.h, g.beta)
        y2 = zr @ Cr + zi @ Ci
        np.testing.assert_allclose(y2, y, rtol=0, atol=5e-14)


@pytest.mark.parametrize("name", CASES + NEWMEAN_CASES)
def test_full_loop_matches_reference(name):
    """Replay the recorded white noise through the oracle controller: every
    iteration's actions / costs / elites / mean / std and the executed actions
    must match the reference run (bit-exact indices; floats to 1e-12)."""
    g = Golden(name)
    calls = iter(range(g.n_noise_calls))

    def noise(num):
        zr, zi = g.noise(next(calls))
        assert zr.shape[0] == num
        return zr, zi

    orc = _oracle_for(g, noise)
    orc.beginning_of_rollout()
    it = 0
    for s in range(g.n_steps):
        a = orc.get_action(g.obs[s])
        np.testing.assert_allclose(a, g.executed[s], rtol=1e-12, atol=1e-14)
        if g.new_mean:   # the mean a subclass' compute_new_mean left behind (its last row is NOT the kept one)
            np.testing.assert_allclose(orc.mean, g.mean_after[s], rtol=1e-12, atol=1e-14)
            assert np.abs(orc.mean[-1] - orc.trace[s][-1].mean[-1]).max() > 1e-3
        for tr in orc.trace[s]:
            ref = g.it(it)
            assert tr.actions.shape == ref["simact"].shape
            np.testing.assert_allclose(tr.actions, ref["simact"], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(tr.costs, ref["costs"], rtol=1e-12, atol=1e-13)
            assert np.array_equal(tr.elite_idx, ref["elite"])
            assert tr.best_idx == ref["best"]
            np.testing.assert_allclose(tr.mean, ref["mean"], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(tr.std, ref["std"], rtol=1e-12, atol=1e-14)
            it += 1
    assert it == g.n_iters_total
    assert next(calls, None) is None


def test_noise_call_order_and_sizes():
    """batch sizes 128->102->81, +3 shifted elites on steps >= 1."""
    g = Golden("c1_halfcheetah_n128")
    sizes = [g.noise(i)[0].shape[0] for i in range(g.n_noise_calls)]
    assert sizes == [128, 102, 81, 128, 3, 102, 81, 128, 3, 102, 81]
    assert O.population_sizes(128, 10, 1.25, 3) == [128, 102, 81]
    assert O.population_sizes(4096, 10, 1.25, 5) == [4096, 3276, 2620, 2096, 1676]
    assert [g.it(i)["simact"].shape[0] for i in range(6)] == [128, 102, 81, 131, 102, 81]
    assert [g.it(i)["costs"].shape[0] for i in range(6)] == [128, 105, 84, 131, 105, 84]


def test_legacy_stream_reproduces_reference():
    """Under np.random.seed the oracle's legacy draw order reproduces the reference run."""
    g = Golden("c1_halfcheetah_n128")
    np.random.seed(g.seed)
    orc = _oracle_for(g, lambda num: O.legacy_white_noise(num, g.d, g.h))
    orc.beginning_of_rollout()
    for s in range(g.n_steps):
        a = orc.get_action(g.obs[s])
        np.testing.assert_allclose(a, g.executed[s], rtol=1e-12, atol=1e-14)


def test_cost_functions_match_reference():
    z = np.load(os.path.join(GOLDEN, "cost_fn_vectors.npz"))
    hc17 = O.CostSpec.halfcheetah(17)
    hc18 = O.CostSpec.halfcheetah(18)
    assert np.array_equal(hc17(z["o17"], z["a6"]), z["hc17"])
    assert np.array_equal(hc18(z["o18"], z["a6"]), z["hc18"])
    assert np.array_equal(O.CostSpec.halfcheetah(17, False)(z["o17"], z["a6"]), z["hc17_noflip"])
    assert np.array_equal(hc17(z["o17"][0, 0][None], z["a6"][0, 0][None])[0], z["hc17_single"])
    assert np.array_equal(O.CostSpec.humanoid_standup()(z["o378"], z["a17"]), z["hs"])
    with pytest.raises(ValueError):
        O.CostSpec.halfcheetah(16)
    # the flip penalty is exercised by the vectors
    assert (np.abs(z["o17"][..., 1]) > np.pi / 2).any()


def env_cost_specs():
    """(fixture tag, oracle spec, has next_obs) for every cost function in env_cost_vectors.npz."""
    return [
        ("ant", O.CostSpec.ant(), True), ("hopper", O.CostSpec.hopper(), True),
        ("humanoid_excl", O.CostSpec.humanoid(24, True), False), ("humanoid_incl", O.CostSpec.humanoid(24, False), False),
        ("reacher", O.CostSpec.reacher(11), False),
        ("fpp_dense", O.CostSpec.fetch_pick_and_place(25, False, 0.05, False), False),
        ("fpp_dense_shaped", O.CostSpec.fetch_pick_and_place(25, False, 0.05, True), False),
        ("fpp_sparse", O.CostSpec.fetch_pick_and_place(25, True, 0.05, False), False),
        ("fpp_sparse_shaped", O.CostSpec.fetch_pick_and_place(25, True, 0.05, True), False),
        ("freach_dense", O.CostSpec.fetch_reach(10, False, 0.05), False),
        ("freach_sparse", O.CostSpec.fetch_reach(10, True, 0.05), False),
        ("door", O.CostSpec.door(39, 30, 30, True, True), False), ("door_plain", O.CostSpec.door(39, 30, 30, False, False), False),
        ("relocate", O.CostSpec.relocate(39, 36, True), False), ("relocate_plain", O.CostSpec.relocate(39, 36, False), False),
    ]


def env_cost_inputs(z, tag):
    base = tag.split("_")[0] if tag.startswith(("fpp", "freach")) else tag   # fpp / freach variants share inputs
    obs, act = z[base + "_obs"], z[base + "_act"]
    nxt = z[base + "_next"] if base + "_next" in z.files else None
    return obs, act, nxt


@pytest.mark.parametrize("tag,spec,has_next", env_cost_specs(), ids=[t for t, _, _ in env_cost_specs()])
def test_env_cost_functions_match_reference(tag, spec, has_next):
    """f-4: Ant / Hopper / Humanoid / Reacher (environments/mujoco.py:151-171, 205-225, 317-343, 366-368),
    FetchPickAndPlace / FetchReach (environments/robotics.py:150-164, 286-295) and Door / Relocate
    (environments/mjenvs.py:57-78, 155-174) as parametric cost terms: indicator terms exact, floats to 1e-12 (the
    reference adds its terms in an env-specific order)."""
    z = np.load(os.path.join(GOLDEN, "env_cost_vectors.npz"))
    obs, act, nxt = env_cost_inputs(z, tag)
    assert (nxt is not None) == has_next and spec.needs_next_obs == has_next
    got = spec(obs, act, nxt)
    tol = 1e-12 if z[tag].dtype == np.float64 else 1e-7     # the sparse robotics costs come back as float32
    want = z[tag].astype(np.float64)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    if spec.health_idx >= 0:   # both healthy and unhealthy rows, and non-finite observations, are in the vectors
        u = spec.unhealthy(obs)
        assert 0 < u.sum() < u.size and not np.isfinite(obs).all()


@pytest.mark.parametrize("name", __import__("golden_util").CEMSTD_CASES)
def test_cem_std_oracle_matches_reference(name):
    """f-3: the CEM baseline MpcCemStd (truncated normal, icem/controllers/mpc.py:142-327): replaying the recorded
    uniform draws through the oracle reproduces every sampled batch, cost, elite list, distribution and bound of the
    reference run (bit-exact indices; floats to 1e-12) and the executed actions."""
    from golden_util import GoldenCemStd
    g = GoldenCemStd(name)
    calls = iter(range(g.n_calls))
    om, oc = O.SyntheticModel(g.A, g.B, g.kind), O.CostSpec.halfcheetah(g.o)
    orc = O.CemStdOracle(horizon=g.h, num_traj=g.N, opt_iterations=g.iters, elites_size=g.K, alpha=g.alpha,
                         init_std=g.init_std, like_levine=g.like_levine, shift_means=g.shift_means,
                         execute_best_elite=g.execute_best, low=g.low, high=g.high,
                         rollout_cost=lambda ob, ac: O.rollout_costs(om, oc, ob, ac, mode=g.cost_mode),
                         uniforms=lambda num: g.call(next(calls))["u"])
    orc.beginning_of_rollout()
    i = 0
    for s in range(g.n_steps):
        a = orc.get_action(g.obs[s])
        np.testing.assert_allclose(a, g.executed[s], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.mean, g.mean_after[s], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.std, g.std_after[s], rtol=0, atol=1e-12)
        for _ in range(g.iters):
            ref, (act, costs, idx, mean, std, lower, upper) = g.call(i), orc.trace[i]
            np.testing.assert_allclose(act, ref["simact"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(costs, ref["costs"], rtol=1e-12, atol=1e-12)
            assert np.array_equal(idx, ref["elite"])
            np.testing.assert_allclose(mean, ref["mean"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(std, ref["std"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(lower, ref["lower_next"], rtol=1e-12, atol=1e-9)
            np.testing.assert_allclose(upper, ref["upper_next"], rtol=1e-12, atol=1e-9)
            i += 1


@pytest.mark.parametrize("name", __import__("golden_util").RANDOM_CASES)
def test_random_shooting_oracle_matches_reference(name):
    """f-3: MpcRandom (icem/controllers/mpc.py:86-138): replaying the recorded action_space.sample() draws through the
    oracle reproduces every sampled sequence (exactly: same affine map), cost, argmin and executed action."""
    from golden_util import GoldenRandom
    g = GoldenRandom(name)
    om, oc = O.SyntheticModel(g.A, g.B, g.kind), O.CostSpec.halfcheetah(g.o)
    orc = O.RandomShootingOracle(horizon=g.h, num_traj=g.N, freq=g.freq, low=g.low, high=g.high,
                                 rollout_cost=lambda ob, ac: O.rollout_costs(om, oc, ob, ac, mode=g.cost_mode),
                                 uniforms=g.block_uniforms)
    for s in range(g.n_steps):
        st = g.step(s)
        a = orc.get_action(st["obs"])
        assert np.array_equal(orc.actions, st["actions"])
        np.testing.assert_allclose(orc.costs, st["costs"], rtol=1e-12, atol=1e-12)
        assert orc.best == int(st["best"]) and np.array_equal(a, st["executed"])
    # the held blocks: the construction-time action serves `freq` calls, every later draw `freq + 1`
    b = O.piecewise_blocks(0, 3 * g.freq + 5, g.freq)
    assert b[:g.freq].tolist() == [0] * g.freq and b[g.freq:2 * g.freq + 1].tolist() == [1] * (g.freq + 1) and b[2 * g.freq + 1] == 2
