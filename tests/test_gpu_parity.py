"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures captured from the reference.  Tolerances (from BASELINE.json north_star): elite index
sets bit-exact; costs / mean / std within 1e-5 relative in f32; the f64 kernels (the reference's
own type) are held to 1e-10."""
import numpy as np
import pytest
import torch

from golden_util import CASES, Golden
from oracle import icem_oracle as O

pytestmark = pytest.mark.gpu

F64_RTOL, F64_ATOL = 1e-10, 1e-12
F32_RTOL, F32_ATOL = 1e-5, 2e-6


def tol(dtype):
    return (dict(rtol=F64_RTOL, atol=F64_ATOL) if dtype == "f64" else dict(rtol=F32_RTOL, atol=F32_ATOL))


def make_planner(g: Golden, dtype, **over):
    from icem_amd import IcemConfig, IcemPlanner
    kw = dict(horizon=g.h, act_dim=g.d, num_traj=g.N, elites_size=g.K, opt_iters=g.iters, cost_mode=g.cost_mode,
              use_mean_actions=g.use_mean, keep_previous_elites=g.keep, shift_elites=g.shift,
              factor_decrease=g.gamma, alpha=g.alpha, init_std=g.init_std, fraction_reused=g.xi,
              noise_beta=g.beta, dtype=dtype, seed=1234)
    kw.update(over)
    pl = IcemPlanner(IcemConfig(**kw), g.low, g.high)
    pl.set_model(g.kind, g.A, g.B)
    spec = O.CostSpec.halfcheetah(g.o) if g.env_kind == "halfcheetah" else O.CostSpec.humanoid_standup()
    pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
    return pl


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


# ---------------------------------------------------------------------------------------------
# K1 sampling
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", CASES)
def test_sample_clip_external_noise_matches_reference(name, dtype):
    """First sampling call of each golden run: same white draws in -> the reference's actions out."""
    g = Golden(name)
    pl = make_planner(g, dtype)
    zr, zi = g.noise(0)
    mean0 = np.zeros((g.h, g.d)) + (g.high + g.low) / 2
    std0 = np.ones((g.h, g.d)) * (g.high - g.low) / 2 * g.init_std
    got = np_(pl.sample_clip(g.N, mean0, std0, zr, zi))
    ref = g.it(0)["simact"]
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, **tol(dtype))
    np.testing.assert_allclose(got, O.sample_action_sequences(mean0, std0, g.low, g.high, g.beta, zr, zi), **tol(dtype))
    assert got.min() >= g.low.min() - 1e-7 and got.max() <= g.high.max() + 1e-7
    if g.beta >= 2:  # enough mass outside the box that clipping is actually exercised
        assert (got == g.high.max()).any() or (got == g.low.min()).any()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_sample_clip_time_slice_and_row0_mean(dtype):
    g = Golden("c1_halfcheetah_n128")
    pl = make_planner(g, dtype)
    rs = np.random.RandomState(0)
    mean = rs.uniform(-0.3, 0.3, (g.h, g.d))
    std = rs.uniform(0.1, 0.6, (g.h, g.d))
    zr, zi = rs.randn(7, g.d, g.h // 2 + 1), rs.randn(7, g.d, g.h // 2 + 1)
    full = O.sample_action_sequences(mean, std, g.low, g.high, g.beta, zr, zi)
    # icem.py:102: only the last time step is written
    out = torch.full((7, g.h, g.d), 9.0, dtype=pl.dt, device=pl.device)
    pl.sample_clip(7, mean, std, zr, zi, t_begin=g.h - 1, out=out)
    got = np_(out)
    assert (got[:, :-1] == 9.0).all()
    np.testing.assert_allclose(got[:, -1], full[:, -1], **tol(dtype))
    # icem.py:87-88: row 0 <- mean (only where first_index == 0)
    got = np_(pl.sample_clip(7, mean, std, zr, zi, row0_mean=True))
    np.testing.assert_allclose(got[0], mean, **tol(dtype))
    np.testing.assert_allclose(got[1:], full[1:], **tol(dtype))
    got = np_(pl.sample_clip(7, mean, std, zr, zi, row0_mean=True, first_index=7))
    np.testing.assert_allclose(got, full, **tol(dtype))


@pytest.mark.parametrize("rounds", [10, 7])
@pytest.mark.parametrize("h,d", [(30, 6), (12, 6), (13, 4), (30, 17), (64, 3), (2, 1)])
def test_philox_normals_match_oracle(h, d, rounds):
    """Device Philox4x32 + Box-Muller == the oracle's restatement (f64: tight; f32: the
    hardware log2/sin/cos path within 2e-6 absolute of the float32 formula)."""
    from icem_amd import IcemConfig, IcemPlanner
    n, seed, off, first = 37, 0xDEADBEEFCAFE1234, (5 << 32) | 17, 1000
    for dtype, npdt, atol in (("f64", np.float64, 1e-12), ("f32", np.float32, 4e-6)):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=64, dtype=dtype, seed=seed, rng_rounds=rounds),
                         -np.ones(d), np.ones(d))
        zr, zi = pl.philox_normals(n, offset=off, first_index=first)
        ozr, ozi = O.philox_white_noise(seed, off, n, d, h, first_index=first, rounds=rounds, dtype=npdt)
        np.testing.assert_allclose(np_(zr), ozr, rtol=0, atol=atol)
        np.testing.assert_allclose(np_(zi), ozi, rtol=0, atol=atol)
        # sampling from the same counters == oracle sampling from the oracle's normals
        mean = np.zeros((h, d))
        std = 0.5 * np.ones((h, d))
        got = np_(pl.sample_clip(n, mean, std, offset=off, first_index=first))
        ref = O.sample_action_sequences(mean, std, -np.ones(d), np.ones(d), 0.25, ozr.astype(np.float64),
                                        ozi.astype(np.float64))
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 if dtype == "f64" else 1e-5)


def test_philox_known_answers_and_moments():
    """Random123 known-answer vectors pin the oracle's Philox; moments pin the device normals."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = O.philox4x32(*[np.array([c]) for c in ctr], key[0], key[1], 10)
        assert tuple(int(x[0]) for x in got) == out
    from icem_amd import IcemConfig, IcemPlanner
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=64, dtype="f32", seed=7), -np.ones(6), np.ones(6))
    zr, zi = pl.philox_normals(20000, offset=3)
    z = np.concatenate([np_(zr).ravel(), np_(zi)[..., 1:-1].ravel()])
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    assert (np_(zi)[..., 0] == 0).all() and (np_(zi)[..., -1] == 0).all()


def test_colored_noise_statistics():
    """KATs of SURVEY 8(c): unit variance and PSD slope -beta of the sampled sequences."""
    from icem_amd import IcemConfig, IcemPlanner
    h, d, n, beta = 30, 6, 20000, 2.0
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=64, dtype="f32", seed=11, noise_beta=beta),
                     -100 * np.ones(d), 100 * np.ones(d))
    x = np_(pl.sample_clip(n, np.zeros((h, d)), np.ones((h, d)), offset=1))  # [n,h,d], no clipping
    # sigma normalises the non-DC bins to unit variance; the random DC bin adds (s0/(h sigma))^2
    Cr, Ci = O.synthesis_matrices(h, beta)
    var_t = (Cr ** 2 + Ci ** 2).sum(axis=0)
    assert np.allclose(var_t, var_t[0]) and abs(var_t[0] - Cr[0, 0] ** 2 - 1.0) < 1e-12
    assert abs(x.std() - np.sqrt(var_t[0])) < 1e-2
    assert abs((x - x.mean(axis=1, keepdims=True)).var(axis=1).mean() - 1.0) < 2e-2
    psd = (np.abs(np.fft.rfft(x, axis=1)) ** 2).mean(axis=(0, 2))
    k = np.arange(1, h // 2)
    slope = np.polyfit(np.log(k), np.log(psd[1:h // 2]), 1)[0]
    assert abs(slope + beta) < 0.1


# ---------------------------------------------------------------------------------------------
# K2 rollout + cost
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", CASES)
def test_rollout_cost_matches_reference(name, dtype):
    g = Golden(name)
    pl = make_planner(g, dtype)
    model = O.SyntheticModel(g.A, g.B, g.kind)
    for i in (0, g.iters):  # first iteration of MPC steps 0 and 1 (the latter includes shifted elites)
        ref = g.it(i)
        act = ref["simact"]
        costs, obs = pl.rollout_cost(g.obs[i // g.iters], act, return_observations=True)
        n_sim = act.shape[0]
        np.testing.assert_allclose(np_(costs), ref["costs"][:n_sim], **tol(dtype))
        np.testing.assert_allclose(np_(obs), O.rollout_observations(model, g.obs[i // g.iters], act),
                                   rtol=tol(dtype)["rtol"], atol=10 * tol(dtype)["atol"])


@pytest.mark.parametrize("mode", ["sum", "best", "final"])
def test_cost_reduce(mode):
    g = Golden("c1_halfcheetah_n128")
    pl = make_planner(g, "f64", cost_mode=mode)
    x = np.random.RandomState(1).randn(333, g.h)
    ref = {"sum": x.sum(1), "best": x.min(1), "final": x[:, -1]}[mode]
    np.testing.assert_allclose(np_(pl.cost_reduce(x)), ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("o", [3, 8, 16, 17, 18, 24, 29, 32])
@pytest.mark.parametrize("kind", [0, 1])
def test_rollout_cost_obs_dims(o, kind):
    """Every compiled observation width (incl. zero-padded ones) against the oracle."""
    from icem_amd import IcemConfig, IcemPlanner
    h, d, n = 12, 5, 300
    rs = np.random.RandomState(o)
    m = O.SyntheticModel.make(o, d, kind)
    spec = O.CostSpec(0.1, o - 1, -1.0, min(1, o - 1), 10.0, 0.3)
    act = rs.uniform(-1, 1, (n, h, d))
    obs0 = rs.randn(o)
    for dtype in ("f64", "f32"):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=64, dtype=dtype), -np.ones(d), np.ones(d))
        pl.set_model(kind, m.A, m.B)
        pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
        got = np_(pl.rollout_cost(obs0, act))
        ref = O.rollout_costs(m, spec, obs0, act)
        if dtype == "f64":
            np.testing.assert_allclose(got, ref, **tol(dtype))
        else:
            # north_star's 1e-5, relative to the magnitude of the sum a cost is (positive and negative step terms cancel),
            # no absolute floor on top -- the bound of tests/test_gpu_parity_sizes.py::_full_loop
            mag = O.rollout_cost_magnitudes(m, spec, obs0, act)
            err = np.abs(got - ref)
            worst = int(np.argmax(err - 1e-5 * mag))
            assert err[worst] <= 1e-5 * mag[worst], (o, kind, worst, err[worst], mag[worst], ref[worst])
    # wider observations: f32 only (k_rollout_wide.hip), up to 384
    with pytest.raises(Exception, match="UNSUPPORTED"):
        pl.set_model(0, np.eye(400), np.zeros((d, 400)))
    pl64 = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=64, dtype="f64"), -np.ones(d), np.ones(d))
    with pytest.raises(Exception, match="UNSUPPORTED"):
        pl64.set_model(0, np.eye(40), np.zeros((d, 40)))


# ---------------------------------------------------------------------------------------------
# K3 top-k, K4 refit
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,k", [(1, 1), (5, 10), (10, 10), (128, 10), (1024, 10), (1025, 10), (4099, 10),
                                 (65536, 10), (65539, 64), (300000, 10)])
def test_topk_sorted_exact(n, k, dtype):
    g = Golden("c1_halfcheetah_n128")
    pl = make_planner(g, dtype)
    rs = np.random.RandomState(n + k)
    c = rs.randn(n).astype(np.float32 if dtype == "f32" else np.float64)
    if n > 20:  # ties, NaN, infinities
        c[rs.randint(0, n, 8)] = c.min() - 1.0
        c[rs.randint(0, n, 3)] = np.nan
        c[rs.randint(0, n, 2)] = -np.inf
        c[rs.randint(0, n, 2)] = np.inf
    oc, oi = pl.topk_sorted(c, k)
    ref = O.topk_sorted(c.astype(np.float64), k)
    kk = min(k, n)
    assert np.array_equal(oi.cpu().numpy()[:kk], ref[:kk])
    cc = np.where(np.isnan(c), np.inf, c)
    assert np.array_equal(np_(oc)[:kk], cc[ref[:kk]].astype(np.float64))
    if n < k:  # fewer candidates than K: padded with (+inf, INT_MAX)
        assert (oi.cpu().numpy()[n:] == np.iinfo(np.int32).max).all() and np.isinf(np_(oc)[n:]).all()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_gather_refit(dtype):
    g = Golden("c1_halfcheetah_n128")
    pl = make_planner(g, dtype)
    rs = np.random.RandomState(3)
    act = rs.uniform(-1, 1, (500, g.h, g.d))
    idx = rs.permutation(500)[:10]
    mean = rs.randn(g.h, g.d) * 0.1
    std = rs.uniform(0.1, 0.5, (g.h, g.d))
    m, s = pl._t(mean).clone(), pl._t(std).clone()
    el = pl.gather_refit(pl._t(act), torch.as_tensor(idx), m, s)
    rm, rstd = O.refit(act[idx], mean, std, g.alpha)
    np.testing.assert_allclose(np_(el), act[idx], **tol(dtype))
    np.testing.assert_allclose(np_(m), rm, **tol(dtype))
    np.testing.assert_allclose(np_(s), rstd, **tol(dtype))
    pl.shift(m, s)
    rm2 = rm.copy()
    rm2[:-1] = rm[1:]
    np.testing.assert_allclose(np_(m), rm2, **tol(dtype))
    np.testing.assert_allclose(np_(s), np.ones_like(rm) * (g.high - g.low) / 2 * g.init_std, **tol(dtype))


# ---------------------------------------------------------------------------------------------
# the whole loop
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", ["f64", "f32", "f32-exact-tile"])
@pytest.mark.parametrize("name", CASES)
def test_fused_plan_replays_reference_run(name, dtype):
    """Fused device path, fed the reference's own white draws: after every CEM iteration of every
    MPC step the elite costs (sorted), mean and std match the reference run, and so do the executed
    actions.  Elite identity is checked through the elite costs + actions (bit-exact index sets).
    In f32 the generic sampler turns the reference's draws into the pool and the TILE kernels roll it out (Tile16H on the
    16-bit matrix cores by default, Tile16 / the exact-f32 pipe under "f32-exact-tile"), the threshold merges select and
    refit: the throughput path from the rollout on, on the reference's own noise."""
    g = Golden(name)
    exact_tile = dtype == "f32-exact-tile"
    dtype = "f32" if exact_tile else dtype
    pl = make_planner(g, dtype)
    if exact_tile:
        pl.set_tile_arith("f32")
    pl.reset()
    calls = iter(range(g.n_noise_calls))
    it_global = [0]
    t = tol(dtype)

    def noise(num):
        zr, zi = g.noise(next(calls))
        assert zr.shape[0] == num
        return zr, zi

    for s in range(g.n_steps):
        def check(it):
            ref = g.it(it_global[0])
            gi = (pl.mpc_step * g.iters + it + 1) & 1
            ea, ec = np_(pl.elites_actions[gi]), np_(pl.elites_costs[gi])
            ref_c = ref["costs"][ref["elite"]]
            np.testing.assert_allclose(ec, ref_c, **t)
            # elite index set: rows of the reference pool at the reference's elite indices
            sim = ref["simact"]
            for r, e in enumerate(ref["elite"]):
                if e < sim.shape[0]:
                    np.testing.assert_allclose(ea[r], sim[e], **t)
            if it < g.iters - 1:
                np.testing.assert_allclose(np_(pl.mean), ref["mean"], **t)
                np.testing.assert_allclose(np_(pl.std), ref["std"], **t)
            else:  # after the last iteration the mean is already shifted, std reset
                sh = ref["mean"].copy()
                sh[:-1] = ref["mean"][1:]
                np.testing.assert_allclose(np_(pl.mean), sh, **t)
            it_global[0] += 1

        a = np_(pl.plan_step(g.obs[s], noise=noise, on_iteration=check))
        np.testing.assert_allclose(a, g.executed[s], **t)
        ref_last = g.it(it_global[0] - 1)
        np.testing.assert_allclose(np_(pl.best_cost)[0], ref_last["costs"].min(), **t)
    assert next(calls, None) is None


@pytest.mark.parametrize("name", ["c1_halfcheetah_n128", "h13_odd_n48", "final_noreuse_n40"])
def test_controller_device_path_legacy_stream(name):
    """MpcICemHip under np.random.seed draws the reference's stream in the reference's order."""
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env
    g = Golden(name)
    env = halfcheetah_env(g.o)
    env.action_space.low[:] = -g.bounds
    env.action_space.high[:] = g.bounds
    if g.d != 6:
        from icem_amd.envs import Box
        env.action_space = Box(-g.bounds * np.ones(g.d), g.bounds * np.ones(g.d))
    ctrl = MpcICemHip(env=env, forward_model=DeviceSyntheticModel(g.A, g.B, g.kind), horizon=g.h,
                      num_simulated_trajectories=g.N, factor_decrease_num=g.gamma,
                      cost_along_trajectory=g.cost_mode, dtype="f64", noise_source="numpy_legacy",
                      action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters,
                                                 init_std=g.init_std, use_mean_actions=g.use_mean,
                                                 keep_previous_elites=g.keep, shift_elites_over_time=g.shift,
                                                 fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert ctrl.device_path and ctrl.has_state and not ctrl.needs_data
    with pytest.raises(AttributeError):
        ctrl.get_action(g.obs[0], None)
    np.random.seed(g.seed)
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    for s in range(g.n_steps):
        a = ctrl.get_action(g.obs[s], None)
        assert a.dtype == np.float64 and a.shape == (g.d,)
        np.testing.assert_allclose(a, g.executed[s], rtol=F64_RTOL, atol=F64_ATOL)
    ctrl.end_of_rollout(1.0, 0.0, "train")
    assert len(ctrl.elite_samples) == ctrl.num_elites


@pytest.mark.parametrize("name", ["c1_halfcheetah_n128", "h12_n64"])
def test_controller_host_model_path(name):
    """A reference-style CPU forward model (predict / predict_n_steps) behind the same controller:
    sampling, top-k and refit on the GPU, model + env.cost_fn on the host."""
    from icem_amd import MpcICemHip, halfcheetah_env
    from icem_amd.models import ForwardModel
    g = Golden(name)
    env = halfcheetah_env(g.o)
    om = O.SyntheticModel(g.A, g.B, g.kind)

    class HostModel(ForwardModel):
        def predict(self, *, observations, states, actions):
            return om.predict(observations, actions), None, np.zeros(observations.shape[:-1] + (1,))

    ctrl = MpcICemHip(env=env, forward_model=HostModel(env=env), horizon=g.h, num_simulated_trajectories=g.N,
                      factor_decrease_num=g.gamma, cost_along_trajectory=g.cost_mode, dtype="f64",
                      noise_source="numpy_legacy",
                      action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters,
                                                 init_std=g.init_std, use_mean_actions=g.use_mean,
                                                 keep_previous_elites=g.keep, shift_elites_over_time=g.shift,
                                                 fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert not ctrl.device_path
    np.random.seed(g.seed)
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    for s in range(g.n_steps):
        a = ctrl.get_action(g.obs[s], None)
        np.testing.assert_allclose(a, g.executed[s], rtol=F64_RTOL, atol=F64_ATOL)


@pytest.mark.parametrize("path,dtype", [("device", "f64"), ("device", "f32"), ("host", "f64")])
def test_compute_new_mean_override_replays_the_reference_subclass(path, dtype):
    """icem.py:168-171, 191-192: a subclass of MpcICem may set the last row of the shifted mean from the best trajectory's last
    predicted observation.  The fixture is the run of such a subclass of the REFERENCE's class (tests/golden/make_golden.py::
    NewMeanICem); the same override on MpcICemHip -- device path (the device epilogue keeps the last row; the override's row
    is written behind it, the one observation it needs re-rolled on demand) and host-model path -- returns the reference's
    actions and leaves the reference's mean behind every step."""
    from golden_util import NEWMEAN_CASES, new_mean_rule
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env
    from icem_amd.envs import Box
    from icem_amd.models import ForwardModel
    g = Golden(NEWMEAN_CASES[0])
    assert g.new_mean
    env = halfcheetah_env(g.o)
    env.action_space = Box(-g.bounds * np.ones(g.d), g.bounds * np.ones(g.d))
    om = O.SyntheticModel(g.A, g.B, g.kind)
    seen = []

    class HostModel(ForwardModel):
        def predict(self, *, observations, states, actions):
            return om.predict(observations, actions), None, np.zeros(observations.shape[:-1] + (1,))

    class NewMean(MpcICemHip):
        def compute_new_mean(self, obs):
            seen.append(np.array(obs))
            return new_mean_rule(obs, self.mean[-1])

    model = DeviceSyntheticModel(g.A, g.B, g.kind) if path == "device" else HostModel(env=env)
    ctrl = NewMean(env=env, forward_model=model, horizon=g.h, num_simulated_trajectories=g.N, factor_decrease_num=g.gamma,
                   cost_along_trajectory=g.cost_mode, dtype=dtype, noise_source="numpy_legacy",
                   action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters, init_std=g.init_std,
                                              use_mean_actions=g.use_mean, keep_previous_elites=g.keep,
                                              shift_elites_over_time=g.shift, fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert ctrl.device_path == (path == "device")
    t = tol(dtype)
    np.random.seed(g.seed)
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    for s in range(g.n_steps):
        a = ctrl.get_action(g.obs[s], None)
        np.testing.assert_allclose(a, g.executed[s], **t)
        np.testing.assert_allclose(ctrl.mean, g.mean_after[s], **t)
    assert len(seen) == g.n_steps and all(ob.shape == (g.o,) for ob in seen)
    # the default keeps the last row: a controller that does not override never asks for the observation
    plain = MpcICemHip(env=env, forward_model=DeviceSyntheticModel(g.A, g.B, g.kind), horizon=g.h, num_simulated_trajectories=g.N,
                       factor_decrease_num=g.gamma, cost_along_trajectory=g.cost_mode, dtype="f64", noise_source="numpy_legacy",
                       action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters, init_std=g.init_std,
                                                  use_mean_actions=g.use_mean, keep_previous_elites=g.keep,
                                                  shift_elites_over_time=g.shift, fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert not plain._new_mean_is_overridden(MpcICemHip)
    np.random.seed(g.seed)
    plain.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    np.testing.assert_allclose(plain.get_action(g.obs[0], None), g.executed[0], **tol("f64"))   # (step 0: same draws, same action)
    assert np.abs(plain.mean[-1] - g.mean_after[0][-1]).max() > 1e-3                              # ... but the kept row behind it


class _SerialGtStyleModel:
    """A host forward model whose ``predict_n_steps`` walks the policy the way the reference's ``GroundTruthModel`` does
    (gt_model.py:76-102): one start state after the other, ``policy.get_action(obs[o], None)`` h times each, one
    rollout per trajectory -- the ROW-WISE branch of ``OpenLoopPolicy``.  Built inside the tests from the icem_amd
    ``ForwardModel`` base."""

    @staticmethod
    def make(env, om, workers=0):
        from icem_amd.models import ForwardModel, TrajectoryBatch

        class Model(ForwardModel):
            calls = []

            def predict(self, *, observations, states, actions):
                return om.predict(observations, actions), None, np.zeros(np.shape(observations)[:-1] + (1,))

            def _serial(self, start_observations, start_states, policy, horizon):
                obs_l, nxt_l, act_l = [], [], []
                for start_obs, _state in zip(start_observations, start_states):
                    o, ro, rn, ra = start_obs, [], [], []
                    for _ in range(horizon):
                        a = policy.get_action(o, None)
                        assert a.ndim == 1
                        n = om.predict(o, a)
                        ro.append(o), rn.append(n), ra.append(a)
                        o = n
                    obs_l.append(ro), nxt_l.append(rn), act_l.append(ra)
                return np.asarray(obs_l), np.asarray(nxt_l), np.asarray(act_l)

            def predict_n_steps(self, *, start_observations, start_states, policy, horizon):
                if start_observations.ndim != 2:
                    raise AttributeError("call predict_n_steps with a batch of states")
                n = start_observations.shape[0]
                self.calls.append(n)
                if not workers:
                    o, nx, a = self._serial(start_observations, start_states, policy, horizon)
                else:
                    # gt_par_model.py:77-94: array_split chunks, one sub-policy per worker, results chained in order
                    chunks = [c for c in np.array_split(range(n), workers) if len(c) > 0]
                    policies = [policy.get_parallel_policy_copy(c) for c in chunks]
                    parts = [self._serial(start_observations[c], [start_states[i] for i in c], sub, horizon)
                             for c, sub in zip(chunks, policies)]
                    o, nx, a = (np.concatenate([p[j] for p in parts]) for j in range(3))
                return TrajectoryBatch(observations=o, next_observations=nx, actions=a,
                                       rewards=np.zeros(a.shape[:2] + (1,))), [None] * n
        return Model(env=env)


@pytest.mark.parametrize("workers", [0, 5])
@pytest.mark.parametrize("name", ["c1_halfcheetah_n128", "h13_odd_n48"])
def test_controller_rowwise_host_models(name, workers):
    """The reference's shipped forward models consume the controller's policy one trajectory at a time
    (``GroundTruthModel``) or as per-worker sub-policies (``ParallelGroundTruthModel``); every settings/*.json goes
    through one of them.  MpcICemHip behind such a model, fed the reference's draws, returns the reference's actions."""
    from icem_amd import MpcICemHip, halfcheetah_env
    g = Golden(name)
    env = halfcheetah_env(g.o)
    if g.d != 6:
        from icem_amd.envs import Box
        env.action_space = Box(-g.bounds * np.ones(g.d), g.bounds * np.ones(g.d))
    model = _SerialGtStyleModel.make(env, O.SyntheticModel(g.A, g.B, g.kind), workers)
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=g.h, num_simulated_trajectories=g.N,
                      factor_decrease_num=g.gamma, cost_along_trajectory=g.cost_mode, dtype="f64",
                      noise_source="numpy_legacy",
                      action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters,
                                                 init_std=g.init_std, use_mean_actions=g.use_mean,
                                                 keep_previous_elites=g.keep, shift_elites_over_time=g.shift,
                                                 fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert not ctrl.device_path
    np.random.seed(g.seed)
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    for s in range(g.n_steps):
        a = ctrl.get_action(g.obs[s], None)
        assert a.dtype == np.float64 and a.shape == (g.d,)
        np.testing.assert_allclose(a, g.executed[s], rtol=F64_RTOL, atol=F64_ATOL)
    # population bookkeeping seen by the model: N (+3 shifted elites from the 2nd step on), then the decayed sizes
    sizes = [g.N]
    for _ in range(g.iters - 1):
        sizes.append(max(2 * g.K, int(sizes[-1] / g.gamma)))
    reuse = int(g.K * g.xi) if g.shift else 0
    want = []
    for s in range(g.n_steps):
        want += [sizes[0] + (reuse if s > 0 else 0)] + sizes[1:]
    assert type(model).calls == want


@pytest.mark.parametrize("on_device", [True, False])
def test_controller_env_reward_as_cost(on_device):
    """use_env_reward_as_cost (abstract_controller.py:76-77: costs = -rewards of the rollouts).  A synthetic model that
    carries the controller's own environment reports that environment's reward = -cost_fn, so the device path serves
    it with the same kernels; a model carrying another environment object goes through the host-model path, scored by
    -rewards.  Both reproduce the reference's recorded actions."""
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env
    g = Golden("c1_halfcheetah_n128")
    env = halfcheetah_env(g.o)
    model_env = env if on_device else halfcheetah_env(g.o)
    ctrl = MpcICemHip(env=env, forward_model=DeviceSyntheticModel(g.A, g.B, g.kind, env=model_env), horizon=g.h,
                      num_simulated_trajectories=g.N, factor_decrease_num=g.gamma, cost_along_trajectory=g.cost_mode,
                      dtype="f64", noise_source="numpy_legacy", use_env_reward_as_cost=True,
                      action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters,
                                                 init_std=g.init_std, use_mean_actions=g.use_mean,
                                                 keep_previous_elites=g.keep, shift_elites_over_time=g.shift,
                                                 fraction_elites_reused=g.xi, noise_beta=g.beta))
    assert ctrl.device_path == on_device
    np.random.seed(g.seed)
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    for s in range(g.n_steps):
        np.testing.assert_allclose(ctrl.get_action(g.obs[s], None), g.executed[s], rtol=F64_RTOL, atol=F64_ATOL)


@pytest.mark.parametrize("deferral", [False, True])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_philox_plan_matches_oracle_and_is_shard_invariant(dtype, deferral):
    """Philox mode end to end: (a) world=1 fused step == oracle driven by the same counters;
    (b) world=2 and world=3, emulated on one GPU by two/three planners whose records are
    concatenated in place of the all-gather, reproduce world=1 (RNG keyed by the global index) -- also with
    icem_set_merge_deferral on, where every merge but the last of a step runs in the next launch's prologue."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)
    seed, iters, N, h, d = 99, 4, 1000, 30, 6
    spec = env.cost_spec

    def mk(rank, world):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype=dtype, seed=seed,
                                    rank=rank, world=world), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
        pl.reset()
        return pl

    obs_seq = [0.1 * np.random.RandomState(s).randn(17) for s in range(3)]
    single = mk(0, 1)
    acts1 = [np_(single.plan_step(o)).copy() for o in obs_seq]
    mean1 = np_(single.mean)

    # (a) oracle
    npdt = np.float64 if dtype == "f64" else np.float32
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=npdt)
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64),
                       lambda ob, ac: O.rollout_costs(om, oc, ob, ac), lambda num: tuple(
                           z.astype(np.float64) for z in noise(num)))
    orc.beginning_of_rollout()
    t = dict(rtol=1e-9, atol=1e-11) if dtype == "f64" else dict(rtol=2e-4, atol=2e-5)
    for s, o in enumerate(obs_seq):
        if s:
            noise.begin_step()
        np.testing.assert_allclose(acts1[s], orc.get_action(o), **t)
    np.testing.assert_allclose(mean1, orc.mean, **t)

    # (b) shard invariance
    import ctypes as C
    from icem_amd import _lib as L
    for world in (2, 3, 8):
        pls = [mk(r, world) for r in range(world)]
        st = pls[0]._stream()
        for pl in pls:  # deferral: non-last merges ride in the next icem_plan_iter_local launch (f32 fast path only)
            L.check(pl.lib.icem_set_merge_deferral(pl._h, int(deferral)))
        for s, o in enumerate(obs_seq):
            for pl in pls:
                pl.obs0.copy_(torch.as_tensor(o, dtype=pl.dt))
            for it in range(iters):
                for pl in pls:
                    L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), s, it, st))
                K = pls[0].K
                full = torch.cat([pl.records[r * K:(r + 1) * K] for r, pl in enumerate(pls)], dim=0)
                for pl in pls:
                    pl.records.copy_(full)  # stands in for the RCCL all-gather
                    L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), s, it, st))
            for pl in pls:
                # every rank holds the same replicated result, identical to the single-GPU run
                assert np.array_equal(np_(pl.executed), acts1[s])
        for pl in pls:
            assert np.array_equal(np_(pl.mean), mean1)


def test_large_population_properties():
    """BASELINE sizes (N=65536, h=30, d=6): size-independent checks of the fused f32 path:
    device top-k == lexsort of the device costs (bit-exact); oracle costs of the device's own
    actions within 1e-5 relative; refit of the gathered elites within 1e-5."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    N, h, d = 65536, 30, 6
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=1, dtype="f32", seed=5,
                                use_mean_actions=False), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    obs = 0.1 * np.random.RandomState(0).randn(17)
    mean0, std0 = np_(pl.mean), np_(pl.std)
    pl.plan_step(obs)
    act = np_(pl.actions[:N])
    costs = np_(pl.costs[:N])
    assert act.min() >= -1 and act.max() <= 1
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    ref_costs = O.rollout_costs(om, oc, obs, act)
    np.testing.assert_allclose(costs, ref_costs, rtol=1e-5, atol=2e-5)
    idx = O.topk_sorted(pl.costs[:N].cpu().numpy(), pl.K)
    ea, ec = pl.current_elites()
    assert np.array_equal(np_(ec), costs[idx])
    assert np.array_equal(np_(ea), act[idx])           # bit-exact elite index set
    assert set(idx) == set(O.topk_sorted(ref_costs, pl.K))  # and the same set as the f64 oracle's costs
    rm, rs_ = O.refit(act[idx], mean0, std0, 0.1)
    sh = rm.copy()
    sh[:-1] = rm[1:]
    np.testing.assert_allclose(np_(pl.mean), sh, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np_(pl.executed), act[idx[0], 0], rtol=0, atol=0)


def _sharded_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = halfcheetah_env(17)
        model = DeviceSyntheticModel.make(17, 6)
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=2000, opt_iters=3, dtype="f32", seed=21,
                                    rank=rank, world=world), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        c = env.cost_spec
        pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
        pl.reset()
        acts = []
        for s in range(3):
            acts.append(np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy())
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), acts=np.array(acts), mean=np_(pl.mean))
    finally:
        dist.destroy_process_group()


def test_sharded_processes_on_one_gpu(tmp_path):
    """The real multi-process path (IcemPlanner.plan_step with rank/world + the all-gather through
    torch.distributed) with two processes sharing this GPU over gloo: every rank ends with the
    single-process result, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=2000, opt_iters=3, dtype="f32", seed=21),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    acts = np.array([np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy() for s in range(3)])
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["acts"], acts)
        assert np.array_equal(z["mean"], np_(pl.mean))


@pytest.mark.parametrize("h,d,o", [(30, 6, 17), (30, 6, 18), (12, 6, 17), (13, 4, 17), (30, 17, 24)])
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("mode", ["sum", "best", "final"])
def test_fast_kernels_all_compiled_shapes(h, d, o, kind, mode):
    """The f32 throughput kernels (folded sampler, matrix-pipe rollout with per-workgroup top-K, one-wave
    merge) on every compiled shape, both model kinds, all three cost reductions and ragged populations
    (not multiples of 64 / of the sampler tile), against the oracle driven by the same RNG stream."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    from icem_amd.envs import CostSpec
    N, seed, iters = 777, 31 + h + o, 3
    low, high = -np.ones(d), np.ones(d)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    spec = O.CostSpec(0.1, o - 2, -1.0, 3, 10.0, 0.05)  # lin/flip on non-default columns: exercises the permutation
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, cost_mode=mode, dtype="f32",
                                seed=seed), low, high)
    pl.set_model(kind, model.A, model.B)
    pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
    rs = np.random.RandomState(5)
    obs0 = 0.3 * rs.randn(o)
    om = O.SyntheticModel(model.A, model.B, kind)
    # K1 + K2 stand-alone
    mean, std = rs.uniform(-0.2, 0.2, (h, d)), rs.uniform(0.2, 0.6, (h, d))
    act = np_(pl.sample_clip(N, mean, std, offset=9, first_index=5))
    zr, zi = O.philox_white_noise(seed, 9, N, d, h, first_index=5, dtype=np.float32)
    ref_act = O.sample_action_sequences(mean, std, low, high, 0.25, zr.astype(np.float64), zi.astype(np.float64))
    np.testing.assert_allclose(act, ref_act, rtol=0, atol=2e-5)
    costs = np_(pl.rollout_cost(obs0, act))
    ref_costs = O.rollout_costs(om, spec, obs0, act, mode=mode)
    np.testing.assert_allclose(costs, ref_costs, rtol=2e-5, atol=5e-5)
    assert (np.abs(O.rollout_observations(om, obs0, act)[..., 3]) > 0.05).any()  # flip term exercised
    # whole MPC steps through the fused plan path
    pl.reset()
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float32)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters,
                                    cost_along_trajectory=mode), low, high,
                       lambda ob, ac: O.rollout_costs(om, spec, ob, ac, mode=mode),
                       lambda num: tuple(z.astype(np.float64) for z in noise(num)))
    orc.beginning_of_rollout()
    for s in range(2):
        if s:
            noise.begin_step()
        a = np_(pl.plan_step(obs0))
        np.testing.assert_allclose(a, orc.get_action(obs0), rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(np_(pl.mean), orc.mean, rtol=5e-4, atol=5e-5)


def test_torch_model_path_matches_oracle():
    """f-2: a torch nn.Module dynamics (MLP, f32) behind MpcICemHip: sampling / cost reduction / top-k /
    refit in HIP, the model's batched steps in torch on the GPU; checked against the oracle driving a NumPy
    copy of the same network with the same RNG stream.  Also runs the module in bf16 (matrix cores) and
    checks self-consistency of the planner's outputs."""
    from icem_amd import MpcICemHip, TorchForwardModel, halfcheetah_env
    torch.manual_seed(0)
    o, d, h, N, iters, seed = 17, 6, 12, 512, 3, 77

    class Dyn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1 = torch.nn.Linear(o + d, 64)
            self.l2 = torch.nn.Linear(64, o)

        def forward(self, obs, act):
            return obs + 0.1 * self.l2(torch.tanh(self.l1(torch.cat([obs, act], dim=-1))))

    net = Dyn()
    cost_t = lambda ob, ac: 0.1 * (ac ** 2).sum(-1) - ob[:, 8] + 10.0 * (ob[:, 1].abs() > np.pi / 2)
    env = halfcheetah_env(o)
    model = TorchForwardModel(net, cost_t, o, d)
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                      cost_along_trajectory="sum", dtype="f32", seed=seed,
                      action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5,
                                                 use_mean_actions=True, keep_previous_elites=True,
                                                 shift_elites_over_time=True, fraction_elites_reused=0.3,
                                                 noise_beta=0.25))
    assert ctrl.torch_path and not ctrl.device_path
    W1, b1 = net.l1.weight.detach().cpu().numpy().astype(np.float64), net.l1.bias.detach().cpu().numpy().astype(np.float64)
    W2, b2 = net.l2.weight.detach().cpu().numpy().astype(np.float64), net.l2.bias.detach().cpu().numpy().astype(np.float64)

    def rollout_cost(obs, actions):
        ob = np.broadcast_to(obs, (actions.shape[0], o)).copy()
        acc = np.zeros(actions.shape[0])
        for t in range(h):
            a = actions[:, t]
            acc += 0.1 * (a ** 2).sum(-1) - ob[:, 8] + 10.0 * (np.abs(ob[:, 1]) > np.pi / 2)
            ob = ob + 0.1 * (np.tanh(np.concatenate([ob, a], -1) @ W1.T + b1) @ W2.T + b2)
        return acc

    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float32)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64), rollout_cost,
                       lambda num: tuple(z.astype(np.float64) for z in noise(num)))
    obs = 0.1 * np.random.RandomState(3).randn(o)
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    orc.beginning_of_rollout()
    for s in range(2):
        if s:
            noise.begin_step()
        a = ctrl.get_action(obs, None)
        np.testing.assert_allclose(a, orc.get_action(obs), rtol=2e-4, atol=2e-5)
    # bf16 module: runs, stays in bounds, deterministic
    model16 = TorchForwardModel(Dyn().to(torch.bfloat16), cost_t, o, d, dtype=torch.bfloat16)
    def mk16(replay):
        return MpcICemHip(env=env, forward_model=model16, horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                          cost_along_trajectory="sum", dtype="f32", seed=seed, deterministic_replay=replay,
                          action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5,
                                                     use_mean_actions=True, keep_previous_elites=True,
                                                     shift_elites_over_time=True, fraction_elites_reused=0.3,
                                                     noise_beta=0.25))
    for replay in (True, False):
        c16 = mk16(replay)
        outs = []
        for rep in range(2):
            c16.beginning_of_rollout(observation=obs, state=None, mode="train")
            outs.append(c16.get_action(obs, None))
        assert np.all(np.abs(outs[0]) <= 1.0) and np.all(np.abs(outs[1]) <= 1.0)
        # deterministic_replay: every episode replays episode 0's noise; default: the noise streams run on across
        # episodes like the reference's np.random stream (icem_set_episode)
        assert np.array_equal(outs[0], outs[1]) == replay


@pytest.mark.parametrize("N,K,iters,keep,shift,use_mean", [
    (2, 10, 2, True, True, True),       # minimum population: K clamps to 2 (icem.py:237-240)
    (20, 10, 3, True, True, True),      # N < one 64-lane tile; decay floor 2*elites_size
    (65, 10, 3, False, True, False),    # one lane into the second tile
    (300, 32, 2, True, False, True),    # largest K of the fast path
    (300, 40, 2, True, True, True),     # K > 32: generic kernels take over
    (1000, 3, 1, True, True, True),     # single iteration per step
])
def test_fast_path_edge_populations(N, K, iters, keep, shift, use_mean):
    """Ragged / tiny populations and K extremes through the whole planner (f32, Philox) vs the oracle."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    seed, h, d = 1000 + N + K, 30, 6
    cfg = IcemConfig(horizon=h, act_dim=d, num_traj=N, elites_size=K, opt_iters=iters, dtype="f32", seed=seed,
                     keep_previous_elites=keep, shift_elites=shift, use_mean_actions=use_mean)
    pl = IcemPlanner(cfg, env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight, c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    pl.reset()
    assert pl.K == max(2, min(K, N // 2))
    p = O.IcemParams(horizon=h, num_simulated_trajectories=N, elites_size=K, opt_iterations=iters,
                     keep_previous_elites=keep, shift_elites_over_time=shift, use_mean_actions=use_mean)
    n_reuse = int(p.num_elites * 0.3)
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, shift=shift and n_reuse > 0, dtype=np.float32)
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    orc = O.IcemOracle(p, env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64),
                       lambda ob, ac: O.rollout_costs(om, oc, ob, ac), lambda num: tuple(
                           z.astype(np.float64) for z in noise(num)) if num > 0 else (np.zeros((0, d, h // 2 + 1)),) * 2)
    orc.beginning_of_rollout()
    rs = np.random.RandomState(N)
    for s in range(3):
        obs = 0.2 * rs.randn(17)
        if s:
            noise.begin_step()
        a = np_(pl.plan_step(obs))
        np.testing.assert_allclose(a, orc.get_action(obs), rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(np_(pl.best_cost)[0], orc.last_min_cost, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(np_(pl.std), orc.std, rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("h,d,o,kind,mode,N", [(30, 6, 18, 1, "best", 16453), (12, 6, 17, 0, "final", 5001),
                                               (13, 4, 17, 1, "sum", 12003), (30, 6, 17, 0, "sum", 33001),
                                               (30, 17, 24, 1, "sum", 5001), (30, 17, 24, 0, "best", 16389)])
def test_many_tiles_per_workgroup(h, d, o, kind, mode, N):
    """More 16-trajectory tiles than candidate lists: workgroups hold 2, 4 or 8 rollout waves (single-launch kernel)
    or 16 (N = 33001: sampler + rollout kernels), the last pass is ragged and the workgroup list merge runs.
    Samples bit-equal to the stand-alone sampler's, costs and the sorted top-K against the oracle."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    low, high = -np.ones(d), np.ones(d)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    spec = O.CostSpec(0.1, 5, -1.0, 5, 10.0, 0.05)  # lin and flip on the same column
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=1, cost_mode=mode, dtype="f32", seed=3,
                                use_mean_actions=False), low, high)
    pl.set_model(kind, model.A, model.B)
    pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
    pl.reset()
    obs0 = 0.3 * np.random.RandomState(1).randn(o)
    mean0, std0 = pl.mean.clone(), pl.std.clone()
    pl.plan_step(obs0)
    act = np_(pl.actions[:N])
    costs = np_(pl.costs[:N])
    assert np.array_equal(act, np_(pl.sample_clip(N, mean0, std0, offset=0)))
    om = O.SyntheticModel(model.A, model.B, kind)
    ref = O.rollout_costs(om, spec, obs0, act, mode=mode)
    np.testing.assert_allclose(costs, ref, rtol=2e-5, atol=5e-5)
    assert np.array_equal(costs, np_(pl.rollout_cost(obs0, pl.actions[:N])))  # same bits as the stand-alone rollout
    idx = O.topk_sorted(pl.costs[:N].cpu().numpy(), pl.K)
    ea, ec = pl.current_elites()
    assert np.array_equal(np_(ec), costs[idx]) and np.array_equal(np_(ea), act[idx])


@pytest.mark.parametrize("h,d,o,kind,N,iters", [(30, 6, 17, 0, 777, 4), (30, 6, 18, 1, 5001, 3), (13, 4, 17, 1, 12003, 4),
                                                (12, 6, 17, 0, 4096, 5), (30, 6, 17, 0, 20011, 3),
                                                (30, 6, 17, 0, 65536, 4), (13, 4, 17, 1, 50001, 3),
                                                (30, 17, 24, 1, 3001, 3), (30, 17, 24, 0, 16384, 3)])
def test_plan_step_merge_prologue_equals_split_api(h, d, o, kind, N, iters):
    """icem_plan_step folds every merge but the last into the next iteration's launch (1, 2 or 4 rollout waves per
    workgroup; N = 20011 mixes 8-wave launches without and 4-wave launches with the prologue; N >= 50001 uses the
    sampler kernel with the prologue followed by the rollout kernel) and ping-pongs the pool
    between the caller's buffer and its own; the split API (icem_plan_iter_local / icem_plan_iter_merge, one merge
    launch per iteration) must give the same bits: distribution, elites, executed action and last pool over 3 MPC steps."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    low, high = -np.ones(d), np.ones(d)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    spec = O.CostSpec(0.1, 3, -1.0, 1, 10.0, 0.5)
    pls = []
    for _ in range(2):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=11), low, high)
        pl.set_model(kind, model.A, model.B)
        pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
        pl.reset()
        pls.append(pl)
    rs = np.random.RandomState(2)
    for step in range(3):
        obs = 0.2 * rs.randn(o)
        a0 = np_(pls[0].plan_step(obs))
        a1 = np_(pls[1].plan_step(obs, on_iteration=lambda it: None))
        assert np.array_equal(a0, a1), step
        for name in ("mean", "std", "best_cost"):
            assert np.array_equal(np_(getattr(pls[0], name)), np_(getattr(pls[1], name))), (step, name)
        (ea0, ec0), (ea1, ec1) = pls[0].current_elites(), pls[1].current_elites()
        assert np.array_equal(np_(ea0), np_(ea1)) and np.array_equal(np_(ec0), np_(ec1)), step
        n_last = pls[0].population_sizes[-1]
        assert np.array_equal(np_(pls[0].actions[:n_last]), np_(pls[1].actions[:n_last])), step
        assert np.array_equal(np_(pls[0].costs[:n_last]), np_(pls[1].costs[:n_last])), step


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_white_noise_branch_philox_plan_matches_oracle(dtype):
    """noise_beta = 0 (np.random.randn(N, h, d), icem.py:77) with the device RNG: draw t of row (n, j) is the sample
    at step t.  Whole MPC steps (generic kernels in f64; single-launch + merge prologue kernels in f32, shifted and
    kept elites) against the oracle driven by the same Philox counters."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=0)
    seed, iters, N, h, d = 5, 3, 600, 30, 6
    spec = env.cost_spec
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, noise_beta=0.0, dtype=dtype, seed=seed),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
    pl.reset()
    npdt = np.float64 if dtype == "f64" else np.float32
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=npdt, white=True)
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters, noise_beta=0.0),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64),
                       lambda ob, ac: O.rollout_costs(om, oc, ob, ac),
                       lambda num: (noise(num)[0].astype(np.float64), None))
    orc.beginning_of_rollout()
    t = dict(rtol=1e-9, atol=1e-11) if dtype == "f64" else dict(rtol=2e-4, atol=2e-5)
    for s in range(3):
        obs = 0.1 * np.random.RandomState(s).randn(17)
        if s:
            noise.begin_step()
        np.testing.assert_allclose(np_(pl.plan_step(obs)), orc.get_action(obs), **t)
    np.testing.assert_allclose(np_(pl.mean), orc.mean, **t)
    # the stand-alone sampler agrees with the oracle's white draw for the same stream
    z = O.philox_white_randn(seed, 7, 50, d, h, dtype=npdt)
    ref = O.sample_action_sequences(orc.mean, orc.std, env.action_space.low, env.action_space.high, 0.0, z.astype(np.float64), None)
    np.testing.assert_allclose(np_(pl.sample_clip(50, orc.mean, orc.std, offset=7)), ref, **tol(dtype))


@pytest.mark.parametrize("N,world", [(140000, 2), (40000, 4)])
def test_sharded_deferred_merge_large_shards(N, world):
    """Shards of 70 000 (sampler + rollout kernels, merge of the gathered records in the sampler's prologue) and of
    10 000 rows (single-launch kernel with the records merge in its prologue), ranks emulated on one GPU, deferral
    on: same executed actions and distribution as the single-GPU icem_plan_step, bit for bit."""
    import ctypes as C
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    from icem_amd import _lib as L
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    spec = env.cost_spec
    iters = 3

    def mk(rank, w):
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=iters, dtype="f32", seed=21, rank=rank, world=w),
                         env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
        pl.reset()
        return pl

    obs_seq = [0.1 * np.random.RandomState(s).randn(17) for s in range(2)]
    single = mk(0, 1)
    acts1 = [np_(single.plan_step(o)).copy() for o in obs_seq]
    pls = [mk(r, world) for r in range(world)]
    st = pls[0]._stream()
    for pl in pls:
        L.check(pl.lib.icem_set_merge_deferral(pl._h, 1))
    K = pls[0].K
    pls[0].profile_enable(True)
    for s, o in enumerate(obs_seq):
        for pl in pls:
            pl.obs0.copy_(torch.as_tensor(o, dtype=pl.dt))
        for it in range(iters):
            for pl in pls:
                L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), s, it, st))
            full = torch.cat([pl.records[r * K:(r + 1) * K] for r, pl in enumerate(pls)], dim=0)
            for pl in pls:
                pl.records.copy_(full)
                L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), s, it, st))
        for pl in pls:
            assert np.array_equal(np_(pl.executed), acts1[s])
    for pl in pls:
        assert np.array_equal(np_(pl.mean), np_(single.mean)) and np.array_equal(np_(pl.std), np_(single.std))
    torch.cuda.synchronize()
    prof = pls[0].profile_read()
    assert prof["merge_refit"][1] == len(obs_seq), prof  # only the last merge of each MPC step was a launch of its own


# ---------------------------------------------------------------------------------------------
# f-3: MpcCemStd (truncated-normal CEM baseline) on the same kernels
# ---------------------------------------------------------------------------------------------
from golden_util import CEMSTD_CASES, GoldenCemStd  # noqa: E402


def _cemstd_controller(g, dtype, noise_source):
    from icem_amd import DeviceSyntheticModel, MpcCemStdHip
    from icem_amd.envs import SyntheticEnv, CostSpec
    import math
    spec = O.CostSpec.halfcheetah(g.o)
    env = SyntheticEnv("HalfCheetah", g.o, g.low.copy(), g.high.copy(),
                       CostSpec(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh))
    model = DeviceSyntheticModel(g.A, g.B, g.kind)
    return MpcCemStdHip(env=env, forward_model=model, horizon=g.h, num_simulated_trajectories=g.N,
                        cost_along_trajectory=g.cost_mode, verbose=False, dtype=dtype, noise_source=noise_source,
                        action_sampler_params=dict(alpha=g.alpha, elites_size=g.K, opt_iterations=g.iters, init_std=g.init_std,
                                                   shift_means=g.shift_means, execute_best_elite=g.execute_best,
                                                   bounds_like_levine=g.like_levine))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", CEMSTD_CASES)
def test_cem_std_controller_replays_reference_run(name, dtype):
    """MpcCemStdHip fed scipy's recorded uniform draws: every sampled batch, elite set and the executed actions /
    distribution after every MPC step of the reference's MpcCemStd run (elite indices exact)."""
    g = GoldenCemStd(name)
    calls = iter(range(g.n_calls))
    ctrl = _cemstd_controller(g, dtype, lambda num: g.call(next(calls))["u"])
    t = dict(rtol=1e-9, atol=1e-10) if dtype == "f64" else dict(rtol=2e-4, atol=2e-5)
    pl = ctrl.planner
    # the stand-alone sampler on the first call
    c0 = g.call(0)
    mean0 = np.zeros((g.h, g.d)) + (g.high + g.low) / 2
    std0, lo0, hi0 = O.cem_bounds(mean0, np.ones((g.h, g.d)) * (g.high - g.low) / 2 * g.init_std, g.low, g.high, g.like_levine)
    np.testing.assert_allclose(lo0, c0["lower"], rtol=1e-12, atol=1e-12)
    got = np_(pl.sample_truncnorm(g.N, std0 * 0 + mean0, std0, lo0, hi0, c0["u"]))
    np.testing.assert_allclose(got, c0["simact"], **(dict(rtol=0, atol=1e-9) if dtype == "f64" else dict(rtol=0, atol=3e-5)))
    ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
    i = 0
    for s in range(g.n_steps):
        a = ctrl.get_action(g.obs[s], None)
        np.testing.assert_allclose(a, g.executed[s], **t)
        np.testing.assert_allclose(ctrl.mean, g.mean_after[s], **t)
        np.testing.assert_allclose(ctrl.std, g.std_after[s], **t)
        i += g.iters
        ref = g.call(i - 1)
        np.testing.assert_allclose(ctrl.elite_samples.as_array("actions"), ref["simact"][ref["elite"]], **t)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cem_std_device_rng_matches_oracle(dtype):
    """MpcCemStdHip on the device RNG against the oracle fed the same Philox uniforms (2 MPC steps, both bound modes)."""
    g = GoldenCemStd(CEMSTD_CASES[0])
    npdt = np.float64 if dtype == "f64" else np.float32
    for like_levine in (False, True):
        g.like_levine = like_levine
        from icem_amd import MpcCemStdHip  # noqa: F401
        ctrl = _cemstd_controller(g, dtype, "philox")
        ctrl.planner.cfg.seed = 0
        om, oc = O.SyntheticModel(g.A, g.B, g.kind), O.CostSpec.halfcheetah(g.o)
        state = {"call": 0}

        def uniforms(num):
            u = O.philox_uniforms(0, state["call"], num, g.d, g.h, dtype=npdt).astype(np.float64)
            state["call"] += 1
            return u
        orc = O.CemStdOracle(horizon=g.h, num_traj=g.N, opt_iterations=g.iters, elites_size=g.K, alpha=g.alpha,
                             init_std=g.init_std, like_levine=like_levine, shift_means=True, execute_best_elite=True,
                             low=g.low, high=g.high, rollout_cost=lambda ob, ac: O.rollout_costs(om, oc, ob, ac, mode=g.cost_mode),
                             uniforms=uniforms)
        orc.beginning_of_rollout()
        ctrl.beginning_of_rollout(observation=g.obs[0], state=None, mode="train")
        t = dict(rtol=1e-8, atol=1e-9) if dtype == "f64" else dict(rtol=3e-4, atol=3e-5)
        for s in range(2):
            np.testing.assert_allclose(ctrl.get_action(g.obs[s], None), orc.get_action(g.obs[s]), **t)
        np.testing.assert_allclose(ctrl.mean, orc.mean, **t)


def test_best_trajectory_view_matches_oracle_rollout():
    """f-1: what the reference hands to visualize_plan / hooks (simulated_paths[best_traj_idx], icem.py:180-183) is
    re-rolled on demand from the best elite's actions: observations / next_observations / actions / cost of that one
    trajectory against the oracle; the elite view keeps the RolloutBuffer-style accessors."""
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=30, num_simulated_trajectories=600, factor_decrease_num=1.25,
                      cost_along_trajectory="sum", verbose=False, dtype="f32", seed=4,
                      action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=3, init_std=0.5, use_mean_actions=True,
                                                 keep_previous_elites=True, shift_elites_over_time=True,
                                                 fraction_elites_reused=0.3, noise_beta=0.25))
    obs = 0.1 * np.random.RandomState(3).randn(17)
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    assert len(ctrl.best_trajectory(obs)) == 0 and len(ctrl.elite_samples) == 0
    a = ctrl.get_action(obs, None)
    bt = ctrl.best_trajectory(obs)
    es = ctrl.elite_samples
    assert len(bt) == 1 and len(es) == 10
    acts = bt.as_array("actions")
    assert np.array_equal(acts[0], es.as_array("actions")[0]) and np.allclose(acts[0, 0], a)
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    ref_obs = O.rollout_observations(om, obs, acts)
    np.testing.assert_allclose(bt.as_array("observations"), ref_obs, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bt.as_array("next_observations")[0, :-1], ref_obs[0, 1:], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bt.as_array("next_observations")[0, -1], om.predict(ref_obs[:, -1], acts[:, -1])[0], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bt.as_array("costs"), O.rollout_costs(om, oc, obs, acts), rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(es.as_array("costs")[0], bt.as_array("costs")[0], rtol=2e-5, atol=5e-5)
    r0 = bt[0]  # per-trajectory dict view, as the reference's consumers index a RolloutBuffer
    assert r0["observations"].shape == (30, 17) and r0["actions"].shape == (30, 6)
    ctrl.do_visualize_plan = "last"   # no live-rendering environment here: must be a no-op, like the reference's guard
    ctrl.get_action(obs, None)


def test_verbose_mode_prints_the_reference_lines_and_plans_the_same(capsys):
    """verbose=True (icem.py:112-115, 151-158): the mean's first row, then one line per CEM iteration with best / mean /
    worst cost (per step for "sum") and the best first action -- from the device buffers, through the per-iteration form
    of the step -- and the executed actions are the quiet controller's, bit for bit."""
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env
    kw = dict(horizon=30, num_simulated_trajectories=300, factor_decrease_num=1.25, cost_along_trajectory="sum", dtype="f32",
              seed=4, action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=3, init_std=0.5, use_mean_actions=True,
                                                 keep_previous_elites=True, shift_elites_over_time=True,
                                                 fraction_elites_reused=0.3, noise_beta=0.25))
    obs = 0.1 * np.random.RandomState(3).randn(17)
    out = {}
    for verbose in (False, True):
        ctrl = MpcICemHip(env=halfcheetah_env(17), forward_model=DeviceSyntheticModel.make(17, 6, kind=1), verbose=verbose, **kw)
        ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
        out[verbose] = [ctrl.get_action(obs, None) for _ in range(2)]
    text = capsys.readouterr().out
    lines = [ln for ln in text.splitlines() if ln.startswith("iter ")]
    assert len(lines) == 6 and lines[0].startswith("iter 0:300 --- best cost:") and lines[1].startswith("iter 1:240 ---")
    assert text.count("--------------------") == 2 and "iCEM using" in text
    best = [float(ln.split("best cost:")[1].split("---")[0]) for ln in lines]
    worst = [float(ln.split("worst:")[1].split("best action")[0]) for ln in lines]
    assert all(b <= w for b, w in zip(best, worst))
    ctrl.check_model_consistency()    # a learned / synthetic model: nothing to compare, nothing printed
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("kind,mode,o,N", [(0, "sum", 17, 4096), (1, "best", 18, 1000), (1, "final", 8, 300), (0, "sum", 24, 700)])
def test_generic_path_forms_compute_the_same_bits(dtype, kind, mode, o, N, monkeypatch):
    """The strict-parity path's forms of an iteration -- a trajectory's row of lanes (rollout_cost_rows_kernel) + ONE selection /
    gather / refit launch (select_refit_kernel) against one thread per trajectory + top-K partials, pack, merge (options gk_rollout_thread = 1, gk_select = 0) -- give the same bits in every buffer over three MPC steps: costs, elite sets
    and their costs, mean, std, executed action (same fused multiply-add chains, same key order, same refit)."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    d = 6
    env = halfcheetah_env(17)   # (action space; the cost is set per width below)
    model = DeviceSyntheticModel.make(o, d, kind=kind)

    def run(form):
        from icem_amd import _lib as L
        L.reset_options()
        if form == "round4":      # one thread per trajectory; top-K partials -> pack -> merge
            L.set_option("gk_rollout_thread", 1)
            L.set_option("gk_select", 0)
        if dtype == "f32":
            L.set_option("disable_fast", 1)   # f32 on the generic kernels
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=d, num_traj=N, opt_iters=3, dtype=dtype, seed=5, cost_mode=mode),
                         env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost(0.1, min(8, o - 1), -1.0, 1, 10.0, 0.3)
        pl.reset()
        out = []
        for s in range(3):
            obs = 0.1 * np.random.RandomState(40 + s).randn(o)
            a = pl.plan_step(obs).cpu().numpy().copy()
            ea, ec = pl.current_elites()
            out.append((a, pl.costs.cpu().numpy().copy(), ea.cpu().numpy().copy(), ec.cpu().numpy().copy(),
                        pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy(), pl.best_cost.cpu().numpy().copy()))
        return out
    old = run("round4")
    for form in ("rows",):
        new = run(form)
        for s, (x, y) in enumerate(zip(new, old)):
            for k, (u, v) in enumerate(zip(x, y)):
                assert np.array_equal(u, v, equal_nan=True), (form, s, k)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_one_launch_selection_with_every_cost_tied(dtype, monkeypatch):
    """A cost that is the same for every trajectory (all weights zero: 4096 keys tie with the threshold, more than the
    selection's candidate array holds): np.argsort's order among equal costs is by index, and the one-launch selection falls
    back to the deterministic rounds -- the elites are pool rows 0 .. K - 1, the kept elites behind them, exactly as the
    three-launch path picks them, step after step."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=0)

    def run(old):
        from icem_amd import _lib as L
        L.reset_options()
        if old:
            L.set_option("gk_select", 0)
        if dtype == "f32":
            L.set_option("disable_fast", 1)
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=4096, opt_iters=3, dtype=dtype, seed=9),
                         env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost(0.0, 0, 0.0, -1, 0.0, 0.0)
        pl.reset()
        out = []
        for s in range(2):
            a = pl.plan_step(0.1 * np.random.RandomState(s).randn(17)).cpu().numpy().copy()
            ea, ec = pl.current_elites()
            out.append((a, ea.cpu().numpy().copy(), ec.cpu().numpy().copy(), pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy()))
        return out, pl
    new, pl = run(False)
    old, _ = run(True)
    assert np.all(new[-1][2] == 0.0)
    assert np.array_equal(new[-1][1], pl.actions[:pl.K].cpu().numpy())     # the last pool's first K rows
    for s, (x, y) in enumerate(zip(new, old)):
        for k, (u, v) in enumerate(zip(x, y)):
            assert np.array_equal(u, v), (s, k)


@pytest.mark.parametrize("h,d,o,K,N,kind,mode", [
    (40, 9, 8, 20, 300, 1, "sum"),      # h > 32: the 64-column synthesis table; d > 8: B through LDS; K > 16: the looped gather
    (50, 3, 24, 4, 90, 0, "best"),      # 32 lanes per trajectory row, a short elite list
    (33, 12, 32, 17, 257, 1, "final"),  # odd horizon, the widest generic observation, ragged last workgroup
])
def test_strict_parity_path_at_unusual_shapes_against_the_oracle(h, d, o, K, N, kind, mode):
    """The float64 path's kernels off the beaten shapes -- long horizons (HMAX = 64 in both samplers), more action dims than
    the rollout keeps in registers, elite counts beyond the register-gathered 16 -- over two MPC steps of device (Philox)
    noise against the NumPy oracle restating the same stream: executed action, mean and std to 1e-9 (icem.py:106-211)."""
    from icem_amd import IcemConfig, IcemPlanner
    iters, seed = 2, 17
    rs = np.random.RandomState(h + d)
    m = O.SyntheticModel.make(o, d, kind)
    spec = O.CostSpec(0.1, o - 1, -1.0, min(1, o - 1), 10.0, 0.3)
    low, high = -0.7 * np.ones(d), 0.9 * np.ones(d)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, elites_size=K, opt_iters=iters, dtype="f64", seed=seed,
                                cost_mode=mode), low, high)
    pl.set_model(kind, m.A, m.B)
    pl.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
    pl.reset()
    sched = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float64)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters, elites_size=K), low, high,
                       lambda ob, ac: O.rollout_costs(m, spec, ob, ac, mode=mode),
                       lambda num: tuple(z.astype(np.float64) for z in sched(num)))
    orc.beginning_of_rollout()
    for step in range(2):
        obs = 0.2 * rs.randn(o)
        if step:
            sched.begin_step()
        got = np_(pl.plan_step(obs))
        want = orc.get_action(obs)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(np_(pl.mean), orc.mean, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(np_(pl.std), orc.std, rtol=1e-9, atol=1e-11)


# ---------------------------------------------------------------------------------------------
# f-4: the remaining env cost functions as device cost terms
# ---------------------------------------------------------------------------------------------

def _env_cost_cases():
    from test_oracle_golden import env_cost_specs
    return env_cost_specs()


def _device_spec(spec):
    """oracle CostSpec -> the product's CostSpec (same fields)."""
    from icem_amd.envs import CostSpec, CostTerm
    import dataclasses
    d = dataclasses.asdict(spec)
    d["terms"] = tuple(CostTerm(**t) for t in d["terms"])
    return CostSpec(**d)


@pytest.mark.parametrize("mode", ["sum", "best", "final"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag,spec,has_next", _env_cost_cases(), ids=[t for t, _, _ in _env_cost_cases()])
def test_trajectory_cost_matches_reference_vectors(tag, spec, has_next, dtype, mode):
    """icem_trajectory_cost (trajectory_cost_fn on rollouts an external model left on the device) with the Ant, Hopper,
    Humanoid, Reacher and Fetch cost terms, on the vectors recorded from the reference's cost functions: f64 against the
    reference's per-step costs reduced over the horizon, f32 against the oracle in f32.  Both memory layouts
    ([n, h, o] and a transposed view of a step-major [h, n, o] buffer)."""
    import os
    from golden_util import GOLDEN
    from test_oracle_golden import env_cost_inputs
    from icem_amd import IcemConfig, IcemPlanner
    z = np.load(os.path.join(GOLDEN, "env_cost_vectors.npz"))
    obs, act, nxt = env_cost_inputs(z, tag)
    n, h = 8, 6
    obs, act = obs.reshape(n, h, -1), act.reshape(n, h, -1)
    nxt = None if nxt is None else nxt.reshape(n, h, -1)
    step = z[tag].astype(np.float64).reshape(n, h)
    d = act.shape[-1]
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=n, elites_size=2, opt_iters=1, cost_mode=mode, dtype=dtype),
                     -np.ones(d), np.ones(d))
    pl.set_cost_spec(_device_spec(spec))
    if dtype == "f64":
        want = {"sum": step.sum(1), "best": step.min(1), "final": step[:, -1]}[mode]
        t = dict(rtol=1e-12, atol=1e-12) if z[tag].dtype == np.float64 else dict(rtol=1e-7, atol=1e-7)
    else:
        want = O.spec_trajectory_costs(spec, obs, act, nxt, mode=mode, dtype=np.float32).astype(np.float64)
        t = dict(rtol=1e-6, atol=1e-6)
        # the f32 run agrees with the reference too, except where an f32-rounded value crosses a threshold
        ref = {"sum": step.sum(1), "best": step.min(1), "final": step[:, -1]}[mode]
        assert np.mean(np.abs(want - ref) <= 1e-4 * (1 + np.abs(ref))) > 0.9
    to = lambda x: torch.as_tensor(x, dtype=pl.dt, device=pl.device)  # noqa: E731
    o_t, a_t = to(obs), to(act)
    n_t = None if nxt is None else to(nxt)
    np.testing.assert_allclose(np_(pl.trajectory_cost(o_t, a_t, n_t)), want, **t)
    # step-major storage, as the torch-model path of the controller keeps it
    o_sm = o_t.transpose(0, 1).contiguous().transpose(0, 1)
    n_sm = None if n_t is None else n_t.transpose(0, 1).contiguous().transpose(0, 1)
    assert o_sm.stride() == (o_t.shape[2], n * o_t.shape[2], 1)
    np.testing.assert_allclose(np_(pl.trajectory_cost(o_sm, a_t, n_sm)), want, **t)


def test_trajectory_cost_rejects_bad_arguments():
    from icem_amd import IcemConfig, IcemPlanner, IcemError
    from icem_amd.envs import ant_env, reacher_env
    pl = IcemPlanner(IcemConfig(horizon=4, act_dim=8, num_traj=8, elites_size=2, opt_iters=1), -np.ones(8), np.ones(8))
    pl.set_cost_spec(ant_env().cost_spec)
    obs = torch.zeros((3, 4, 113), dtype=pl.dt, device=pl.device)
    act = torch.zeros((3, 4, 8), dtype=pl.dt, device=pl.device)
    with pytest.raises(IcemError, match="next_observations"):
        pl.trajectory_cost(obs, act, None)
    pl.set_cost_spec(reacher_env(11).cost_spec)
    with pytest.raises(IcemError, match="outside the observation"):
        pl.trajectory_cost(obs[..., :2].contiguous(), act, None)   # the slice [8, 11) does not exist in o=2
    assert np_(pl.trajectory_cost(obs[..., :11].contiguous(), act, None)).tolist() == [0.0] * 3


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("env_name", ["hopper", "reacher", "fetch_sparse"])
def test_rollout_and_plan_with_cost_terms(env_name, dtype):
    """The built-in model + the extra cost terms (general rollout kernel): icem_rollout_cost against the oracle's
    rollout, then whole Philox-driven MPC steps against the oracle loop (elite order decided by those costs)."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    from icem_amd import envs as E
    if env_name == "hopper":
        # healthy range moved to where the synthetic latent lives so that healthy and unhealthy steps both occur
        env = E.hopper_env(healthy_z_range=(-0.05, float("inf")), healthy_state_range=(-0.45, 0.45))
        spec = O.CostSpec.hopper(healthy_z_range=(-0.05, float("inf")), healthy_state_range=(-0.45, 0.45))
    elif env_name == "reacher":
        env, spec = E.reacher_env(11), O.CostSpec.reacher(11)
    else:
        env, spec = E.fetch_reach_env(10, sparse=True, threshold=0.12), O.CostSpec.fetch_reach(10, True, 0.12)
    o, d = env.obs_dim, env.action_space.shape[0]
    h, N, iters, seed = 12, 600, 3, 5
    model = DeviceSyntheticModel.make(o, d, kind=1)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype=dtype, seed=seed),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    npdt = np.float64 if dtype == "f64" else np.float32
    om = O.SyntheticModel(model.A, model.B, model.kind)
    rs = np.random.RandomState(3)
    obs0 = 0.2 * rs.randn(o)
    acts = rs.uniform(-1, 1, (257, h, d))
    got = np_(pl.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)))
    want = O.rollout_costs(om, spec, obs0, acts, dtype=npdt).astype(np.float64)
    if dtype == "f64":
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    else:   # an f32 tanh one ulp off can move a step across a threshold: allow a few whole-penalty mismatches
        close = np.abs(got - want) <= 1e-4 * (1 + np.abs(want))
        assert close.mean() > 0.98
    if env_name == "hopper":   # trajectories with no, some and many unhealthy steps are all in the batch
        import dataclasses
        count = O.rollout_costs(om, dataclasses.replace(spec, diff_idx=-1, ctrl_weight=0.0, health_penalty=1.0), obs0, acts)
        assert (count == 0).any() and (count >= h // 2).any() and 0.1 < count.mean() / h < 0.9
    if dtype == "f32":
        return
    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=npdt)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64),
                       lambda ob, ac: O.rollout_costs(om, spec, ob, ac), lambda num: tuple(
                           zz.astype(np.float64) for zz in noise(num)))
    orc.beginning_of_rollout()
    for s in range(2):
        ob = 0.2 * np.random.RandomState(10 + s).randn(o)
        if s:
            noise.begin_step()
        np.testing.assert_allclose(np_(pl.plan_step(ob)), orc.get_action(ob), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(np_(pl.mean), orc.mean, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("planes", [0, 2])   # icem_set_wide_exact: fp16 planes (default) / bf16 planes
@pytest.mark.parametrize("n", [80, 257, 409, 640])
def test_wide_split_is_the_same_every_launch(n, planes):
    """rollout_wide_split_kernel at o = 376 with the Humanoid cost terms, the same launch 40 times with other kernels in
    between (they leave LDS and registers dirty): every launch returns the bits of the first, and every cost is the
    exact-f32 kernel's to 1e-4 or a whole health penalty away (a tanh one ulp off across a threshold).  The batch
    prologue once zeroed X and stored step 0's actions into it with no barrier between -- different threads, same words:
    whether a trajectory's first actions survived was a matter of timing, in the rows the LAST elements of the action
    block map to (60-63 of a 64-row batch, 60-79 of a five-tile one), and only builds in which wave 0 ran ahead of the
    zeroing waves showed it (EXPERIMENTS R4.9).  n = 80, 257 and 409 leave some workgroup FIVE tiles; 640 is regular
    batches only."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    from icem_amd import envs as E
    env = E.humanoid_env(healthy_z_range=(-0.05, 2.0))
    o, d, h = env.obs_dim, env.action_space.shape[0], 12
    model = DeviceSyntheticModel.make(o, d, kind=1)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=max(n, 64), opt_iters=1, dtype="f32", seed=7),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    rs = np.random.RandomState(3)
    obs0 = 0.2 * rs.randn(o)
    acts = torch.as_tensor(rs.uniform(-1, 1, (n, h, d)) * env.action_space.high, dtype=pl.dt, device=pl.device)
    pl.set_wide_exact(True)
    exact = np_(pl.rollout_cost(obs0, acts)).astype(np.float64)
    pl.set_wide_exact(planes)
    first = pl.rollout_cost(obs0, acts).clone()
    junk = torch.empty(16 << 20, device=pl.device)
    for i in range(40):
        if i % 3 == 0:
            junk.normal_()
        got = pl.rollout_cost(obs0, acts)
        assert torch.equal(got, first), (i, (got != first).nonzero().flatten().tolist()[:8])
    got = np_(first).astype(np.float64)
    far = np.abs(got - exact) > 1e-4 * (1 + np.abs(exact))
    assert far.mean() <= 0.02, (np.nonzero(far)[0][:8], got[far][:4], exact[far][:4])


@pytest.mark.parametrize("env_name", ["ant", "humanoid", "door", "relocate"])
def test_wide_rollout_with_cost_terms(env_name):
    """The built-in model at observation widths 32 < o <= 384 WITH the extra cost terms (k_rollout_wide.hip: the tile kernel
    on the f32 matrix pipe and the row kernel that scores trailing shifted elites): Ant (o = 113: difference term read
    from the NEXT observation, closed health range, finite check over the whole row), Humanoid (o = 376: open health
    range), Door and Relocate (o = 39: norms, gated norms, offsets, step bonuses).  icem_rollout_cost against the float64
    oracle rollout; then whole MPC steps, whose last pool -- tile rows and row-kernel rows -- is re-scored by the oracle."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner
    from icem_amd import envs as E
    if env_name == "ant":   # ranges moved to where the synthetic latent lives: healthy and unhealthy steps both occur
        env, spec = E.ant_env(healthy_z_range=(-0.05, 1.0)), O.CostSpec.ant(healthy_z_range=(-0.05, 1.0))
    elif env_name == "humanoid":
        env, spec = E.humanoid_env(healthy_z_range=(-0.05, 2.0)), O.CostSpec.humanoid(healthy_z_range=(-0.05, 2.0))
    elif env_name == "door":
        env, spec = E.door_env(), O.CostSpec.door()
    else:
        env, spec = E.relocate_env(), O.CostSpec.relocate()
    o, d = env.obs_dim, env.action_space.shape[0]
    assert o > 32
    h, N, iters = 12, 16 * 40, 3
    model = DeviceSyntheticModel.make(o, d, kind=1)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=7),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    om = O.SyntheticModel(model.A, model.B, model.kind)
    rs = np.random.RandomState(3)
    obs0 = 0.2 * rs.randn(o)
    acts = rs.uniform(-1, 1, (257, h, d)) * env.action_space.high
    got = np_(pl.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)))
    want = O.rollout_costs(om, spec, obs0, acts).astype(np.float64)

    def close(a, b):   # an f32 tanh one ulp off can move a step across a threshold: a few whole-penalty mismatches
        return np.abs(a - b) <= 1e-4 * (1 + np.abs(b))
    assert close(got, want).mean() > 0.98, (np.abs(got - want).max(), close(got, want).mean())
    if env_name in ("ant", "humanoid"):   # no, some and many unhealthy steps are all in the batch
        import dataclasses
        count = O.rollout_costs(om, dataclasses.replace(spec, diff_idx=-1, ctrl_weight=0.0, lin_weight=0.0, health_penalty=1.0), obs0, acts)
        assert (count == 0).any() and (count > 0).any()
    if env_name == "door":   # the other reductions over the steps (np.amin / last step), same kernels
        for mode in ("best", "final"):
            pm = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=1, dtype="f32", seed=7, cost_mode=mode),
                             env.action_space.low, env.action_space.high)
            pm.set_model(model.kind, model.A, model.B)
            pm.set_cost_spec(env.cost_spec)
            pm.reset()
            gm = np_(pm.rollout_cost(obs0, torch.as_tensor(acts, dtype=pm.dt, device=pm.device)))
            wm = O.rollout_costs(om, spec, obs0, acts, mode=mode).astype(np.float64)
            assert close(gm, wm).mean() > 0.98, (mode, np.abs(gm - wm).max())
    for s_ in range(2):   # whole MPC steps: the pool of the last iteration holds sampled rows and shifted-elite rows
        ob = 0.2 * np.random.RandomState(20 + s_).randn(o)
        a0 = np_(pl.plan_step(ob))
        assert np.all(np.isfinite(a0))
        n_last = pl.population_sizes[-1]
        pool = np_(pl.actions[:n_last]).astype(np.float64)
        rescored = O.rollout_costs(om, spec, ob.astype(np.float32).astype(np.float64), pool).astype(np.float64)
        dev = np_(pl.costs[:n_last]).astype(np.float64)
        assert close(dev, rescored).mean() > 0.98, (np.abs(dev - rescored).max(), close(dev, rescored).mean())
    # one iteration per MPC step: from the second step on the pool is N sampled rows (whole tiles) + the shifted elites,
    # which rollout_rows_wide_kernel scores row by row
    p1 = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=1, dtype="f32", seed=9),
                     env.action_space.low, env.action_space.high)
    p1.set_model(model.kind, model.A, model.B)
    p1.set_cost_spec(env.cost_spec)
    p1.reset()
    n_shift = int(p1.cfg.num_elites * p1.cfg.fraction_reused)
    assert n_shift >= 1
    for s_ in range(2):
        ob = 0.2 * np.random.RandomState(30 + s_).randn(o)
        p1.plan_step(ob)
    pool = np_(p1.actions[:N + n_shift]).astype(np.float64)
    rescored = O.rollout_costs(om, spec, ob.astype(np.float32).astype(np.float64), pool).astype(np.float64)
    dev = np_(p1.costs[:N + n_shift]).astype(np.float64)
    assert close(dev[N:], rescored[N:]).all() or close(dev, rescored).mean() > 0.995, (dev[N:], rescored[N:])


def test_torch_model_with_env_cost_spec_uses_trajectory_cost():
    """f-2 + f-4: a torch dynamics model without its own cost callable is scored by the env's parametric cost through
    icem_trajectory_cost (Ant-shaped: o=113, difference and health terms); checked against the oracle loop."""
    from icem_amd import MpcICemHip, TorchForwardModel
    from icem_amd.envs import ant_env
    torch.manual_seed(1)
    env = ant_env()
    o, d, h, N, iters, seed = 113, 8, 10, 256, 3, 31

    class Dyn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(o + d, o)

        def forward(self, obs, act):
            return obs + 0.05 * torch.tanh(self.l(torch.cat([obs, act], dim=-1)))

    net = Dyn().double()
    model = TorchForwardModel(net, None, o, d)
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                      cost_along_trajectory="sum", dtype="f64", seed=seed,
                      action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5,
                                                 use_mean_actions=True, keep_previous_elites=True,
                                                 shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=0.25))
    assert ctrl.torch_path and ctrl.torch_spec_cost
    W, b = net.l.weight.detach().cpu().numpy(), net.l.bias.detach().cpu().numpy()
    spec = O.CostSpec.ant()

    def rollout_cost(obs, actions):
        ob = np.broadcast_to(obs, (actions.shape[0], o)).copy()
        obs_all, nxt_all = [], []
        for t in range(h):
            nx = ob + 0.05 * np.tanh(np.concatenate([ob, actions[:, t]], -1) @ W.T + b)
            obs_all.append(ob)
            nxt_all.append(nx)
            ob = nx
        return O.spec_trajectory_costs(spec, np.stack(obs_all, 1), actions, np.stack(nxt_all, 1))

    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float64)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64), rollout_cost,
                       lambda num: tuple(zz.astype(np.float64) for zz in noise(num)))
    orc.beginning_of_rollout()
    ctrl.beginning_of_rollout(observation=np.zeros(o), state=None, mode="train")
    rs = np.random.RandomState(4)
    for s in range(2):
        ob = 0.3 * rs.randn(o)
        ob[2] = 0.25 + 0.1 * s     # z close to the lower end of the healthy range: both outcomes occur in the rollouts
        if s:
            noise.begin_step()
        np.testing.assert_allclose(ctrl.get_action(ob, None), orc.get_action(ob), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N", [4096, 300])
def test_nan_costs_rank_last_with_index_tie_break(N, dtype):
    """A NaN in the start observation makes every cost NaN: the loop must not hang, NaN ranks as +inf and ties
    break by index (icem_topk_sorted's contract, the same on the fast f32 path, the general path and the oracle's
    topk_sorted) -- so the elites are trajectories 0..K-1 of the last pool and the distribution stays finite."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=0)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=3, dtype=dtype, seed=1,
                                keep_previous_elites=False, shift_elites=False), env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    ob = 0.1 * np.random.RandomState(0).randn(17)
    ob[8] = np.nan
    executed = np_(pl.plan_step(ob))
    torch.cuda.synchronize()
    K = pl.K
    assert np.isinf(np_(pl.best_cost)).all() and np.isfinite(np_(pl.mean)).all() and np.isfinite(np_(pl.std)).all()
    last_pool = pl.population_sizes[-1]
    elite_actions, elite_costs = pl.current_elites()
    assert np.isinf(np_(elite_costs)).all()
    assert np.array_equal(np_(elite_actions), np_(pl.actions[:K])) and last_pool >= K
    assert np.array_equal(executed, np_(pl.actions[0, 0]))
    assert O.topk_sorted(np.full(last_pool, np.nan), K).tolist() == list(range(K))


def test_every_cost_tied_in_the_noise_ahead_launch():
    """The same at N = 65 536 (f32): the noise-ahead launch's prologue selects on ONE wave with the lists' first three depths
    in registers (merge_select_shallow) -- with every key tied at +inf all of them survive the threshold: the walk past the
    third depth and, beyond 64 survivors, the tournament of the streaming form; the K elites must be rows 0 .. K-1 again."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=0)
    for keep in (False, True):
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=65536, opt_iters=3, dtype="f32", seed=1,
                                    keep_previous_elites=keep, shift_elites=False), env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        ob = 0.1 * np.random.RandomState(0).randn(17)
        ob[8] = np.nan
        executed = np_(pl.plan_step(ob))
        torch.cuda.synchronize()
        K = pl.K
        assert np.isinf(np_(pl.best_cost)).all() and np.isfinite(np_(pl.mean)).all() and np.isfinite(np_(pl.std)).all()
        elite_actions, elite_costs = pl.current_elites()
        assert np.isinf(np_(elite_costs)).all()
        if not keep:   # (kept elites carry indices behind the pool's: with them the first K pool rows still win the ties)
            assert np.array_equal(np_(elite_actions), np_(pl.actions[:K]))
            assert np.array_equal(executed, np_(pl.actions[0, 0]))


# ---------------------------------------------------------------------------------------------
# f-3: the random-shooting baseline MpcRandom
# ---------------------------------------------------------------------------------------------

def _random_controller(g, dtype, noise_source, seed=0):
    from icem_amd import DeviceSyntheticModel, MpcRandomHip
    from icem_amd.envs import SyntheticEnv
    spec = O.CostSpec.halfcheetah(g.o)
    env = SyntheticEnv("fake", g.o, g.low, g.high, _device_spec(spec))
    model = DeviceSyntheticModel(g.A, g.B, g.kind)
    return MpcRandomHip(env=env, forward_model=model, horizon=g.h, num_simulated_trajectories=g.N,
                        cost_along_trajectory=g.cost_mode, dtype=dtype, seed=seed, noise_source=noise_source,
                        action_sampler_params=dict(action_change_frequency=g.freq))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", __import__("golden_util").RANDOM_CASES)
def test_random_shooting_controller_replays_reference_run(name, dtype):
    """MpcRandomHip fed the recorded action_space.sample() draws reproduces the reference's MpcRandom run
    (icem/controllers/mpc.py:86-138): the sampled sequences (f64: exactly), the argmin and the executed actions; and
    does so too when it draws from the global np.random stream itself in the reference's call order."""
    from golden_util import GoldenRandom
    g = GoldenRandom(name)
    for source in ("callable", "action_space"):
        if source == "action_space":
            np.random.seed(31)   # the seed of the recorded run (make_golden.py::main)
            ctrl = _random_controller(g, dtype, "action_space")
        else:
            ctrl = _random_controller(g, dtype, g.block_uniforms)
        assert ctrl.device_path
        ctrl.beginning_of_rollout(observation=g.step(0)["obs"], state=None, mode="train")
        for s in range(g.n_steps):
            st = g.step(s)
            a = ctrl.get_action(st["obs"], None)
            if dtype == "f64":
                assert np.array_equal(np_(ctrl._last_actions), st["actions"]) and np.array_equal(a, st["executed"])
                np.testing.assert_allclose(np_(ctrl._last_costs), st["costs"], rtol=1e-10, atol=1e-12)
            else:
                np.testing.assert_allclose(np_(ctrl._last_actions), st["actions"], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(a, st["executed"], rtol=1e-6, atol=1e-7)
                np.testing.assert_allclose(np_(ctrl._last_costs), st["costs"], **tol(dtype))
            assert ctrl.best_traj_idx == int(st["best"])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_random_shooting_device_rng_matches_oracle(dtype):
    """Philox mode of the random-shooting sampler: block (b, j) -> word 0 of its stream; oracle driven by the same
    counters.  Populations large enough for the matrix-pipe rollout (f32) and several MPC steps (the call counter and
    the held blocks run on across steps)."""
    from icem_amd import DeviceSyntheticModel, MpcRandomHip, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)
    N, h, freq, seed = 3000, 30, 4, 17
    ctrl = MpcRandomHip(env=env, forward_model=model, horizon=h, num_simulated_trajectories=N, cost_along_trajectory="sum",
                        dtype=dtype, seed=seed, action_sampler_params=dict(action_change_frequency=freq))
    npdt = np.float64 if dtype == "f64" else np.float32
    om, oc = O.SyntheticModel(model.A, model.B, model.kind), O.CostSpec.halfcheetah(17)
    orc = O.RandomShootingOracle(horizon=h, num_traj=N, freq=freq, low=env.action_space.low, high=env.action_space.high,
                                 rollout_cost=lambda ob, ac: O.rollout_costs(om, oc, ob, ac),
                                 uniforms=lambda b0, nb: O.philox_block_uniforms(seed, b0, nb, 6, dtype=npdt))
    ctrl.beginning_of_rollout(observation=np.zeros(17), state=None, mode="train")
    t = dict(rtol=1e-12, atol=1e-13) if dtype == "f64" else dict(rtol=1e-6, atol=1e-6)
    for s in range(3):
        ob = 0.1 * np.random.RandomState(s).randn(17)
        a, want = ctrl.get_action(ob, None), orc.get_action(ob)
        np.testing.assert_allclose(np_(ctrl._last_actions), orc.actions, **t)
        assert ctrl.best_traj_idx == orc.best
        np.testing.assert_allclose(a, want, **t)
        acts = np_(ctrl._last_actions).reshape(-1, 6)
        n_blocks = N * h // (freq + 1)   # distinct held values per step (f32: a handful of the 18 000 draws collide)
        assert (np.abs(acts) <= 1).all() and 0.995 * n_blocks <= len(np.unique(acts[:, 0])) <= n_blocks + 1


# ---------------------------------------------------------------------------------------------
# BASELINE config 5: learned-dynamics rollout (declared RSSM), N=1024, h=12
# ---------------------------------------------------------------------------------------------

def _rssm_numpy(net):
    """Float64 NumPy restatement of icem_amd.models.declared_rssm's module (torch.nn.GRUCell gate order r, z, n);
    cross-checked against oracle/rssm_oracle.py, the port bench.py times as the CPU baseline of --workload c5."""
    from oracle import rssm_oracle as RO
    P = {k: v.detach().cpu().double().numpy() for k, v in net.state_dict().items()}
    _rs = np.random.RandomState(0)
    _o, _a = _rs.randn(5, 230), _rs.uniform(-1, 1, (5, 3, 6))
    _want = RO.rollout_costs(RO.params_from_state_dict(net.state_dict()), _o[0], _a)
    det = net.det
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))  # noqa: E731

    def step(obs, act):
        h, z = obs[:, :det], obs[:, det:]
        x = np.maximum(np.concatenate([z, act], -1) @ P["inp.weight"].T + P["inp.bias"], 0)
        gi = x @ P["gru.weight_ih"].T + P["gru.bias_ih"]
        gh = h @ P["gru.weight_hh"].T + P["gru.bias_hh"]
        r = sig(gi[:, :det] + gh[:, :det])
        u = sig(gi[:, det:2 * det] + gh[:, det:2 * det])
        n = np.tanh(gi[:, 2 * det:] + r * gh[:, 2 * det:])
        h2 = (1 - u) * n + u * h
        z2 = np.maximum(h2 @ P["prior1.weight"].T + P["prior1.bias"], 0) @ P["prior2.weight"].T + P["prior2.bias"]
        return np.concatenate([h2, z2], -1)

    def reward(obs):
        a = np.maximum(obs @ P["rew1.weight"].T + P["rew1.bias"], 0)
        a = np.maximum(a @ P["rew2.weight"].T + P["rew2.bias"], 0)
        return (a @ P["rew3.weight"].T + P["rew3.bias"])[:, 0]

    _ob = np.broadcast_to(_o[0], (5, 230)).copy()
    _acc = np.zeros(5)
    for _t in range(3):
        _acc -= reward(_ob)
        _ob = step(_ob, _a[:, _t])
    np.testing.assert_allclose(_acc, _want, rtol=1e-12, atol=1e-12)
    return step, reward


def test_config5_learned_dynamics_rollout():
    """BASELINE configs[4]: N=1024, h=12, d=6, beta=0.25 with learned dynamics.  The declared RSSM (f32) behind
    MpcICemHip -- sampling, cost reduction, top-K and refit in HIP, the recurrent model's GEMMs in torch on the GPU --
    against the oracle loop driving a float64 NumPy restatement of the same network on the same Philox draws; then the
    same weights in bf16 (matrix cores): costs of a fixed action batch within bf16 accuracy of the f32 ones, and the
    planner deterministic."""
    from icem_amd import MpcICemHip, declared_rssm, halfcheetah_env
    N, h, d, iters, seed = 1024, 12, 6, 3, 5
    env = halfcheetah_env(17)   # action space only (d=6, +-1); the cost is the model's reward head
    asp = dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5, use_mean_actions=True,
               keep_previous_elites=True, shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=0.25)

    def controller(model):
        return MpcICemHip(env=env, forward_model=model, horizon=h, num_simulated_trajectories=N, factor_decrease_num=1.25,
                          cost_along_trajectory="sum", dtype="f32", seed=seed, action_sampler_params=asp)

    m32 = declared_rssm(seed=3)
    ctrl = controller(m32)
    assert ctrl.torch_path and not ctrl.torch_spec_cost
    step, reward = _rssm_numpy(m32.module)

    def rollout_cost(obs, actions):
        ob = np.broadcast_to(obs, (actions.shape[0], obs.shape[0])).copy()
        acc = np.zeros(actions.shape[0])
        for t in range(h):
            acc -= reward(ob)
            ob = step(ob, actions[:, t])
        return acc

    noise = O.PhiloxNoiseSchedule(seed, iters, d, h, dtype=np.float32)
    orc = O.IcemOracle(O.IcemParams(horizon=h, num_simulated_trajectories=N, opt_iterations=iters),
                       env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64), rollout_cost,
                       lambda num: tuple(z.astype(np.float64) for z in noise(num)))
    obs = 0.3 * np.random.RandomState(1).randn(230)
    ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
    orc.beginning_of_rollout()
    for s in range(2):
        if s:
            noise.begin_step()
        np.testing.assert_allclose(ctrl.get_action(obs, None), orc.get_action(obs), rtol=5e-4, atol=5e-5)
    # bf16 weights and activations: same planner, matrix-core GEMMs
    m16 = declared_rssm(seed=3, dtype=torch.bfloat16)
    acts = torch.as_tensor(np.random.RandomState(2).uniform(-1, 1, (N, h, d)), dtype=torch.float32, device="cuda")
    c32 = np_(controller(m32)._costs_of(obs, acts))
    c16_ctrl = controller(m16)
    c16 = np_(c16_ctrl._costs_of(obs, acts))
    assert np.abs(c16 - c32).max() <= 0.05 * (1 + np.abs(c32).max()), (np.abs(c16 - c32).max(), np.abs(c32).max())
    assert np.corrcoef(c16, c32)[0, 1] > 0.99
    outs = []
    c16_ctrl.deterministic_replay = True  # every episode replays episode 0's noise streams
    for _ in range(2):
        c16_ctrl.beginning_of_rollout(observation=obs, state=None, mode="train")
        outs.append(c16_ctrl.get_action(obs, None))
    assert np.array_equal(outs[0], outs[1]) and np.all(np.abs(outs[0]) <= 1.0)


@pytest.mark.parametrize("seed", range(40))
def test_trajectory_cost_random_term_lists(seed):
    """Property test of the cost-term evaluator: random term lists (every kind, gated and not, slices of random place
    and length, plus random flip / linear / difference / health settings) on random rollouts, f64 device against the
    oracle's restatement (1e-12), for the [n, h, o] layout and all three reductions."""
    import dataclasses
    from icem_amd import IcemConfig, IcemPlanner
    rs = np.random.RandomState(100 + seed)
    n, h, o, d = int(rs.randint(1, 40)), int(rs.randint(2, 20)), int(rs.randint(4, 90)), int(rs.randint(1, 9))
    terms = []
    for _ in range(rs.randint(0, 9)):
        kind = int(rs.randint(6))
        ln = 1 if kind in (O.TERM_STEP_GT, O.TERM_SQ_OFFSET) else int(rs.randint(1, min(o, 12) + 1))
        a = int(rs.randint(0, o - ln + 1))
        b = int(rs.randint(0, o - ln + 1)) if (kind in (0, 1, 2) and rs.randint(2)) else -1
        gate = int(rs.randint(o)) if rs.randint(3) == 0 else -1
        terms.append(O.CostTerm(kind, a, b, ln, float(rs.randn()), float(abs(rs.randn())), gate, float(0.3 * rs.randn())))
    health = int(rs.randint(o)) if rs.randint(2) else -1
    spec = O.CostSpec(float(rs.rand()), int(rs.randint(o)), float(rs.randn()) if rs.randint(3) else 0.0,
                      int(rs.randint(o)) if rs.randint(2) else -1, 10.0, float(rs.rand()),
                      diff_idx=int(rs.randint(o)) if rs.randint(2) else -1, diff_weight=float(rs.randn()),
                      health_idx=health, health_penalty=float(50 * rs.rand()), health_lo=-0.5, health_hi=0.8,
                      health_closed=bool(rs.randint(2)), box_from=int(rs.randint(o)) if (health >= 0 and rs.randint(2)) else -1,
                      box_lo=-2.0, box_hi=2.0, terms=tuple(terms))
    obs, nxt, act = rs.randn(n, h, o), rs.randn(n, h, o), rs.uniform(-1, 1, (n, h, d))
    if health >= 0 and n > 2:
        obs[0, 0, rs.randint(o)] = np.inf
        obs[1, h - 1, rs.randint(o)] = np.nan
    for mode in ("sum", "best", "final"):
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=max(n, 2), elites_size=2, opt_iters=1, cost_mode=mode,
                                    dtype="f64"), -np.ones(d), np.ones(d))
        pl.set_cost_spec(_device_spec(spec))
        to = lambda x: torch.as_tensor(x, dtype=pl.dt, device=pl.device)  # noqa: E731
        got = np_(pl.trajectory_cost(to(obs), to(act), to(nxt) if spec.diff_idx >= 0 else None))
        want = O.spec_trajectory_costs(spec, obs, act, nxt, mode=mode)
        finite = np.isfinite(want)
        np.testing.assert_allclose(got[finite], want[finite], rtol=1e-12, atol=1e-12)
        assert np.array_equal(np.isnan(got), np.isnan(want))


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_get_action_host_call_equals_plan_step(dtype):
    """icem_get_action (host observation in, host action out, one synchronisation inside the library) returns exactly
    what icem_plan_step leaves in the device buffers, over several MPC steps; the controller uses it."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, MpcICemHip, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)

    def mk():
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=2000, opt_iters=4, dtype=dtype, seed=9),
                         env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        return pl

    a, b = mk(), mk()
    rs = np.random.RandomState(0)
    for s in range(3):
        ob = 0.1 * rs.randn(17)
        act, best = a.get_action_host(ob)
        want = np_(b.plan_step(ob))
        assert np.array_equal(act, want) and best == float(np_(b.best_cost)[0])
        assert np.array_equal(np_(a.mean), np_(b.mean))
    with pytest.raises(ValueError):
        a.get_action_host(np.zeros(5))
    ctrl = MpcICemHip(env=env, forward_model=model, horizon=30, num_simulated_trajectories=2000, factor_decrease_num=1.25,
                      cost_along_trajectory="sum", dtype=dtype, seed=9,
                      action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=4, init_std=0.5,
                                                 use_mean_actions=True, keep_previous_elites=True,
                                                 shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=0.25))
    c = mk()
    ob = 0.1 * np.random.RandomState(5).randn(17)
    ctrl.beginning_of_rollout(observation=ob, state=None, mode="train")
    assert np.array_equal(ctrl.get_action(ob, None), np_(c.plan_step(ob))) and ctrl.last_min_cost == float(np_(c.best_cost)[0])


@pytest.mark.parametrize("mode,n,h", [("sum", 1007, 12), ("best", 1007, 12), ("final", 1007, 12), ("sum", 8200 + 5, 12),
                                      ("best", 16384 + 21, 12), ("sum", 1, 12), ("final", 17, 1), ("sum", 33, 30),
                                      ("sum", 2048, 12), ("best", 2049 + 30, 12)])
def test_fused_rssm_rollout_kernel(mode, n, h):
    """icem_rssm_rollout_cost (the declared RSSM's whole rollout + reward head in one launch, bf16 MFMA with f32
    accumulation and f32 recurrent state) against (a) a float64 NumPy rollout of the same network with the weights and
    the GEMM inputs rounded to bf16 exactly where the kernel rounds them -- tight; (b) the plain float64 network --
    bf16 accuracy.  Ragged n (not a multiple of the 16-trajectory tile, a single trajectory), horizons 1 / 12 / 30;
    every n here runs the split launch of icem_rssm_split.hip (two tiles per recurrence workgroup above 4096 rows); the fused
    kernel is held to the same costs bit for bit by test_split_rssm_launch_equals_the_fused_kernel."""
    from icem_amd import DeviceRSSMModel
    m = DeviceRSSMModel(seed=3)
    d = 6
    rs = np.random.RandomState(4)
    acts = rs.uniform(-1, 1, (n, h, d))
    obs = 0.3 * rs.randn(230)
    got = np_(m.rollout_cost(obs, torch.as_tensor(acts, dtype=torch.float32, device="cuda"), {"sum": 0, "best": 1, "final": 2}[mode]))

    def bf(x):   # round to bf16 (nearest even), back to float64
        return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()

    P = {k: v.detach().cpu().double().numpy() for k, v in m.reference.state_dict().items()}
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))  # noqa: E731

    def rollout(q):   # q: rounding applied to weights and to every GEMM input (identity = the exact network)
        W = {k: (q(v) if k.endswith("weight") or "weight_" in k else v) for k, v in P.items()}
        hh = np.broadcast_to(obs[:200].astype(np.float32).astype(np.float64), (n, 200)).copy()
        z = np.broadcast_to(obs[200:].astype(np.float32).astype(np.float64), (n, 30)).copy()
        steps = []
        for t in range(h):
            hq, zq, aq = q(hh), q(z), q(acts[:, t])
            a1 = q(np.maximum(np.concatenate([hq, zq], -1) @ W["rew1.weight"].T + P["rew1.bias"], 0))
            a2 = q(np.maximum(a1 @ W["rew2.weight"].T + P["rew2.bias"], 0))
            steps.append(-(a2 @ W["rew3.weight"].T + P["rew3.bias"])[:, 0])
            x = q(np.maximum(np.concatenate([zq, aq], -1) @ W["inp.weight"].T + P["inp.bias"], 0))
            gi = x @ W["gru.weight_ih"].T + P["gru.bias_ih"]
            gh = hq @ W["gru.weight_hh"].T + P["gru.bias_hh"]
            r = sig(gi[:, :200] + gh[:, :200])
            u = sig(gi[:, 200:400] + gh[:, 200:400])
            nn = np.tanh(gi[:, 400:] + r * gh[:, 400:])
            hh = (1 - u) * nn + u * hh
            p = q(np.maximum(q(hh) @ W["prior1.weight"].T + P["prior1.bias"], 0))
            z = p @ W["prior2.weight"].T + P["prior2.bias"]
        s = np.stack(steps, 1)
        return {"sum": s.sum(1), "best": s.min(1), "final": s[:, -1]}[mode]

    emu, exact = rollout(bf), rollout(lambda x: x)
    scale = 1 + np.abs(exact).max()
    assert np.abs(got - emu).max() <= 2e-3 * scale, (np.abs(got - emu).max(), scale)
    assert np.abs(got - exact).max() <= 5e-2 * scale
    if n > 1 and exact.std() > 1e-2 * scale:   # ('best' can pick the shared first step for every trajectory: a constant)
        assert np.corrcoef(got, exact)[0, 1] > 0.99


def test_rssm_staging_can_be_trimmed_and_comes_back():
    """icem_rssm_trim frees the split launch's per-(device, stream) staging areas; the next rollout on that stream
    allocates a new one and returns the same costs (and a second stream gets an area of its own)."""
    from icem_amd import DeviceRSSMModel
    m = DeviceRSSMModel(seed=3)
    rs = np.random.RandomState(8)
    acts = torch.as_tensor(rs.uniform(-1, 1, (300, 12, 6)), dtype=torch.float32, device="cuda")
    obs = 0.3 * rs.randn(230)
    first = np_(m.rollout_cost(obs, acts))
    torch.cuda.synchronize()
    assert m.lib.icem_rssm_trim() == 0
    again = np_(m.rollout_cost(obs, acts))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        other = m.rollout_cost(obs, acts)
    side.synchronize()
    assert np.array_equal(first, again) and np.array_equal(first, np_(other))
    torch.cuda.synchronize()
    assert m.lib.icem_rssm_trim() == 0


def test_split_rssm_launch_equals_the_fused_kernel(tmp_path):
    """Populations up to 65 536 take icem_rssm_split.hip (recurrence workgroups + reward-head workgroups exchanging the
    states through global memory, weights of the head resident in registers); ICEM_RSSM_SPLIT=0 keeps them on the
    fused kernel.  Same arithmetic, same rounding points: the costs must be bit-identical -- ragged tiles, horizons
    1 / 2 / 3 / 5 / 12 / 30, all three cost modes, 128 tiles (both workgroup kinds resident together), 256 (the last size with
    one tile per recurrence workgroup), 257 and up to 4 096 tiles (two tiles per recurrence workgroup, reward workgroups
    walking the tiles behind them), and every case three times in a row (the tile flags
    must be back at zero behind every launch)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "dbg", "rssm_split_check.py")
    f = str(tmp_path / "fused.npz")
    env = dict(os.environ, ICEM_RSSM_SPLIT="0")
    a = subprocess.run([sys.executable, tool, "write", f], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert a.returncode == 0, a.stdout + a.stderr
    env.pop("ICEM_RSSM_SPLIT")
    b = subprocess.run([sys.executable, tool, "check", f], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert b.returncode == 0 and "0 differ" in b.stdout, b.stdout + b.stderr


def test_split_rssm_launch_replays_in_a_hip_graph():
    """The split learned-dynamics launch keeps per-(device, stream) staging and per-tile flags that every launch must
    leave as it found them: captured once (after a first call on the capture stream -- the staging area is allocated
    there, include/icem_hip.h) it is replayed on fresh actions and must give what a direct call gives, bit for bit."""
    from icem_amd import DeviceRSSMModel
    m = DeviceRSSMModel(seed=3)
    n, h = 1000, 12
    obs = 0.3 * np.random.RandomState(1).randn(230)
    acts = torch.empty((n, h, 6), dtype=torch.float32, device="cuda")
    rs = np.random.RandomState(2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        acts.copy_(torch.as_tensor(rs.uniform(-1, 1, (n, h, 6)), dtype=torch.float32))
        m.rollout_cost(obs, acts)          # first call on this stream: allocates the staging area
        side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = m.rollout_cost(obs, acts)
    for rep in range(3):
        fresh = torch.as_tensor(rs.uniform(-1, 1, (n, h, 6)), dtype=torch.float32, device="cuda")
        acts.copy_(fresh)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        direct = m.rollout_cost(obs, fresh)
        torch.cuda.synchronize()
        assert np.array_equal(np_(out), np_(direct)), rep


def test_split_rssm_launches_on_two_streams_at_once():
    """Two split learned-dynamics launches in flight on two streams (each stream has its own staging area and flags; both
    launches' recurrence workgroups are dispatched ahead of their reward workgroups, so neither can starve the other):
    same costs as one at a time, bit for bit -- with both workgroup kinds resident together (n = 1000) and with the
    reward workgroups walking the tiles behind two-tile recurrence workgroups (n = 9000)."""
    from icem_amd import DeviceRSSMModel
    m = DeviceRSSMModel(seed=3)
    obs = 0.3 * np.random.RandomState(1).randn(230)
    rs = np.random.RandomState(5)
    for n in (1000, 9000):
        a1 = torch.as_tensor(rs.uniform(-1, 1, (n, 12, 6)), dtype=torch.float32, device="cuda")
        a2 = torch.as_tensor(rs.uniform(-1, 1, (n, 12, 6)), dtype=torch.float32, device="cuda")
        r1, r2 = m.rollout_cost(obs, a1).clone(), m.rollout_cost(obs, a2).clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(6):
            with torch.cuda.stream(s1):
                o1 = m.rollout_cost(obs, a1)
            with torch.cuda.stream(s2):
                o2 = m.rollout_cost(obs, a2)
            torch.cuda.synchronize()
            assert torch.equal(o1, r1) and torch.equal(o2, r2)


def test_config5_fused_rssm_behind_controller():
    """BASELINE configs[4] with the fused kernel: MpcICemHip + DeviceRSSMModel (one launch per population) against the
    oracle loop driving the bf16-exact... no: the float64 network; the planner's choice must agree up to bf16 cost noise:
    the executed action's cost under the exact network is within bf16 accuracy of the oracle's best."""
    from icem_amd import DeviceRSSMModel, MpcICemHip, declared_rssm, halfcheetah_env
    N, h, d, iters, seed = 1024, 12, 6, 3, 5
    env = halfcheetah_env(17)
    asp = dict(alpha=0.1, elites_size=10, opt_iterations=iters, init_std=0.5, use_mean_actions=True,
               keep_previous_elites=True, shift_elites_over_time=True, fraction_elites_reused=0.3, noise_beta=0.25)
    fused = MpcICemHip(env=env, forward_model=DeviceRSSMModel(seed=3), horizon=h, num_simulated_trajectories=N,
                       factor_decrease_num=1.25, cost_along_trajectory="sum", dtype="f32", seed=seed, action_sampler_params=asp)
    torch32 = MpcICemHip(env=env, forward_model=declared_rssm(seed=3), horizon=h, num_simulated_trajectories=N,
                         factor_decrease_num=1.25, cost_along_trajectory="sum", dtype="f32", seed=seed, action_sampler_params=asp)
    assert fused.rssm_path and not fused.torch_path and torch32.torch_path
    obs = 0.3 * np.random.RandomState(1).randn(230)
    acts = torch.as_tensor(np.random.RandomState(2).uniform(-1, 1, (N, h, d)), dtype=torch.float32, device="cuda")
    cf, ct = np_(fused._costs_of(obs, acts)), np_(torch32._costs_of(obs, acts))
    assert np.abs(cf - ct).max() <= 0.05 * (1 + np.abs(ct).max()) and np.corrcoef(cf, ct)[0, 1] > 0.99
    for c in (fused, torch32):
        c.beginning_of_rollout(observation=obs, state=None, mode="train")
    a_f, a_t = fused.get_action(obs, None), torch32.get_action(obs, None)
    assert np.all(np.abs(a_f) <= 1.0)
    # same Philox draws, costs equal up to bf16 noise: the two planners end on plans of (nearly) the same quality
    assert abs(fused.last_min_cost - torch32.last_min_cost) <= 0.05 * (1 + abs(torch32.last_min_cost))
    again = MpcICemHip(env=env, forward_model=DeviceRSSMModel(seed=3), horizon=h, num_simulated_trajectories=N,
                       factor_decrease_num=1.25, cost_along_trajectory="sum", dtype="f32", seed=seed, action_sampler_params=asp)
    again.beginning_of_rollout(observation=obs, state=None, mode="train")
    assert np.array_equal(again.get_action(obs, None), a_f)   # deterministic


# ---------------------------------------------------------------------------------------------
# in-library elite exchange (icem_exchange_*, csrc/exchange.hip)
# ---------------------------------------------------------------------------------------------

def _xchg_planner(rank, world, dtype, N=1000, iters=4, seed=99, kind=1):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=kind)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=iters, dtype=dtype, seed=seed, rank=rank, world=world),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    return pl


@pytest.mark.parametrize("deferral", [False, True])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("world,N", [(2, 1000), (3, 1000), (8, 1000), (4, 40000), (2, 80000), (16, 300),
                                     # rank 0's shard fills the 256 slabs exactly (1 and 2 tiles per slab): its shifted
                                     # elites of the later MPC steps must not change the launch shape behind the host's back
                                     (2, 8192), (2, 16384), (8, 32768),
                                     # BASELINE configs[3] as written: N = 65 536 global over 8 GPUs, 8 192 rows each
                                     (8, 65536)])
def test_in_library_exchange_emulated_worlds(world, N, dtype, deferral):
    """All ranks of a sharded run as planners of ONE process, connected through the in-library exchange (blocks handed
    over as pointers): the records travel by exchange_push_kernel, the merges wait on the flags of their own block --
    no host copy between icem_plan_iter_local and icem_plan_iter_merge.  Bit-equal to the single-GPU run, with the
    merges as launches of their own and folded into the next local launch (deferral)."""
    import ctypes as C
    from icem_amd import IcemPlanner, _lib as L
    iters = 4
    obs_seq = [0.1 * np.random.RandomState(s).randn(17) for s in range(3)]
    single = _xchg_planner(0, 1, dtype, N, iters)
    want = [np_(single.plan_step(o)).copy() for o in obs_seq]
    pls = [_xchg_planner(r, world, dtype, N, iters) for r in range(world)]
    IcemPlanner.connect_exchange_local(pls)
    st = pls[0]._stream()
    for pl in pls:
        L.check(pl.lib.icem_set_merge_deferral(pl._h, int(deferral)))
    for s, o in enumerate(obs_seq):
        for pl in pls:
            pl.obs0.copy_(torch.as_tensor(o, dtype=pl.dt))
        for it in range(iters):
            for pl in pls:  # local launch + pack + push of every rank ...
                L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), s, it, st))
            for pl in pls:  # ... then the merges (or their stashing): they find every rank's records in their own block
                L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), s, it, st))
        for pl in pls:
            assert np.array_equal(np_(pl.executed), want[s])
    for pl in pls:
        assert np.array_equal(np_(pl.mean), np_(single.mean)) and np.array_equal(np_(pl.std), np_(single.std))
        assert pl.exchange_status()[0] == 0


def _spawn_ranks(fn, world, args, timeout=900):
    """torch.multiprocessing.spawn for ranks that may have to SHARE a GPU (this box has one): every rank of a shared GPU
    gets its own slice of the CUs (HSA_CU_MASK, read when the child's HSA runtime starts).  A rank's launch spins on its
    peers' records; without the slices a peer whose launch finds no free CU meanwhile is only released by the exchange's
    poll budget (seconds) -- a property of two processes on one device, not of the exchange: on a node every rank has a
    GPU of its own and nothing is masked."""
    import multiprocessing as pymp
    import os
    ctx = pymp.get_context("spawn")
    n_dev = max(1, torch.cuda.device_count())
    per_dev = -(-world // n_dev)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    old = os.environ.get("HSA_CU_MASK")
    procs = []
    try:
        for r in range(world):
            if per_dev > 1 and old is None:
                cus = n_cu // per_dev
                os.environ["HSA_CU_MASK"] = f"{r % n_dev}:{(r // n_dev) * cus}-{(r // n_dev + 1) * cus - 1}"
            p = ctx.Process(target=fn, args=(r, world) + tuple(args))
            p.start()
            procs.append(p)
    finally:
        if old is None:
            os.environ.pop("HSA_CU_MASK", None)
    for p in procs:
        p.join(timeout)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _xchg_worker(rank, world, port, out_dir, dtype, N=2000, steps=3):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pl = _xchg_planner(rank, world, dtype, N, 3, seed=21, kind=0)
        pl.connect_exchange()
        acts = []
        for s in range(steps):
            acts.append(np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy())  # icem_plan_step_sharded
        torch.cuda.synchronize()
        status, fine = pl.exchange_status()
        us = pl.exchange_probe(50)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), acts=np.array(acts), mean=np_(pl.mean), status=status, fine=fine, us=us)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,N", [("f32", 2000), ("f64", 2000), ("f32", 12000), ("f32", 70000), ("f32", 8192)])
def test_in_library_exchange_two_processes_ipc(tmp_path, dtype, N):
    """The real multi-process path: two processes share this GPU, exchange their IPC handles once over gloo, and run
    whole MPC steps with icem_plan_step_sharded -- the records move through IPC-mapped peer blocks, the only
    torch.distributed traffic is the handle exchange at construction.  (Sharing one GPU, each process gets half of its CUs:
    _spawn_ranks -- a rank spinning on a peer that cannot get a CU would only ever be released by its poll budget.)
    Every rank ends with the single-process result."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _spawn_ranks(_xchg_worker, 2, (port, str(tmp_path), dtype, N))
    pl = _xchg_planner(0, 1, dtype, N, 3, seed=21, kind=0)
    acts = np.array([np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy() for s in range(3)])
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert int(z["status"]) == 0
        assert np.array_equal(z["acts"], acts)
        assert np.array_equal(z["mean"], np_(pl.mean))
        assert 0 < float(z["us"]) < 1e5


def test_rccl_allgather_elites_on_one_rank():
    """icem_allgather_elites (collective.hip): RCCL bound at run time, a communicator owned by the handle
    (icem_rccl_unique_id -> icem_rccl_connect), ncclAllGather of the K records in place on the launch stream.  With one
    rank the gather must leave the records as they are; a second handle adopts the first one's communicator."""
    import ctypes as C
    from icem_amd import _lib as L
    pl = _xchg_planner(0, 1, "f32", 1000, 2)
    pl._ensure_buffers()
    ident = (C.c_ubyte * L.RCCL_ID_BYTES)()
    L.check(pl.lib.icem_rccl_unique_id(ident))
    assert any(bytes(ident))
    assert b"rccl" in pl.lib.icem_rccl_library()
    L.check(pl.lib.icem_rccl_connect(pl._h, ident))
    pl.plan_step(0.1 * np.random.RandomState(0).randn(17))
    pl.records.copy_(torch.randn_like(pl.records))
    before = pl.records.clone()
    L.check(pl.lib.icem_allgather_elites(pl._h, C.c_void_p(pl.records.data_ptr()), pl._stream()))
    torch.cuda.synchronize()
    assert torch.equal(pl.records, before)
    # without a communicator: a state error, not a crash
    other = _xchg_planner(0, 1, "f32", 1000, 2)
    other._ensure_buffers()
    assert other.lib.icem_allgather_elites(other._h, C.c_void_p(other.records.data_ptr()), other._stream()) == L.ICEM_E_STATE
    L.check(pl.lib.icem_rccl_disconnect(pl._h))
    assert pl.lib.icem_allgather_elites(pl._h, C.c_void_p(pl.records.data_ptr()), pl._stream()) == L.ICEM_E_STATE


def _rccl_worker(rank, world, port, out_dir, N):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if torch.cuda.device_count() >= world:
            torch.cuda.set_device(rank)
        pl = _xchg_planner(rank, world, "f32", N, 3, seed=21, kind=0)
        ok = pl.connect_rccl()
        acts = []
        if ok:
            for s in range(3):
                acts.append(np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy())  # icem_plan_step_sharded over RCCL
            torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), ok=ok, acts=np.array(acts), mean=np_(pl.mean),
                 err=str(getattr(pl, "rccl_error", "")))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_plan_step_sharded_over_rccl_two_processes(tmp_path):
    """The fallback collective inside the library: two ranks, no in-library exchange connected, an RCCL communicator per
    handle -- icem_plan_step_sharded gathers the records with ncclAllGather between pack and merge (one C call per MPC
    step, no host collective) and every rank ends with the single-process result.  RCCL wants one GPU per rank: on a
    one-GPU box the communicator cannot be created and the test says so."""
    import socket
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL wants one GPU per rank and this box has one (two ranks on one device end in 'duplicate GPU' or hang "
                    "inside ncclCommInitRank; IcemPlanner.connect_rccl refuses up front): needs >= 2 GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    N = 12000
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path), N), nprocs=2, join=True)
    z = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    if not all(bool(x["ok"]) for x in z):
        assert torch.cuda.device_count() < 2, [str(x["err"]) for x in z]   # with a GPU per rank it has to work
        pytest.skip("RCCL refuses two ranks on one GPU (" + str(z[0]["err"])[:120] + "): needs >= 2 GPUs")
    pl = _xchg_planner(0, 1, "f32", N, 3, seed=21, kind=0)
    acts = np.array([np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy() for s in range(3)])
    for x in z:
        assert np.array_equal(x["acts"], acts)
        assert np.array_equal(x["mean"], np_(pl.mean))


@pytest.mark.soak
@pytest.mark.parametrize("N", [12000, 70000])
def test_soak_two_processes_ipc_many_steps(tmp_path, N):
    """The two-process run over many MPC steps (ICEM_SOAK_STEPS, default 150): every step of every rank bit-equal to the
    single-process run -- the riding pack, the published merge and their flags / written-through stores under
    repetition (a stale read would show up as one differing step)."""
    import os
    import socket
    import torch.multiprocessing as mp
    steps = int(os.environ.get("ICEM_SOAK_STEPS", "150"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    _spawn_ranks(_xchg_worker, 2, (port, str(tmp_path), "f32", N, steps))
    pl = _xchg_planner(0, 1, "f32", N, 3, seed=21, kind=0)
    acts = np.array([np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy() for s in range(steps)])
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert int(z["status"]) == 0
        bad = np.flatnonzero((z["acts"] != acts).any(axis=1))
        assert bad.size == 0, f"rank {r}: steps {bad[:10]} differ"
        assert np.array_equal(z["mean"], np_(pl.mean))


@pytest.mark.parametrize("n,n_keep,K,special", [(1024, 3, 10, "plain"), (819, 3, 10, "ties"), (300, 0, 10, "nan"), (5, 2, 10, "short"),
                                                 (16381, 3, 32, "plain"), (64, 0, 1, "plain"), (1000, 10, 32, "ties")])
def test_update_distribution_equals_topk_plus_gather_refit(n, n_keep, K, special):
    """icem_update_distribution (top-K over [pool | kept elites] + gather + refit, one launch) against icem_topk_sorted
    over the concatenation + icem_gather_refit: same costs, indices, elites, mean and std, bit for bit -- also with
    ties (broken by index, kept elites behind the pool), NaN / inf costs and fewer candidates than K (padding)."""
    from icem_amd import IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    pl = IcemPlanner(IcemConfig(horizon=12, act_dim=6, num_traj=max(n, 2), opt_iters=1, dtype="f32", seed=1, elites_size=min(K, 32)),
                     env.action_space.low, env.action_space.high)
    rs = np.random.RandomState(n + K)
    costs = rs.randn(n).astype(np.float32)
    kc = rs.randn(n_keep).astype(np.float32)
    if special == "ties":
        costs = np.round(costs * 2) / 2
        kc = np.round(kc * 2) / 2
        if n_keep:
            kc[0] = costs.min()
    if special == "nan":
        costs[::7] = np.nan
        costs[3] = -np.inf
        costs[5] = np.inf
    pool = rs.randn(n, 12, 6).astype(np.float32)
    ka = rs.randn(n_keep, 12, 6).astype(np.float32)
    dev = lambda x: torch.as_tensor(x, device="cuda")  # noqa: E731
    m1, s1 = dev(rs.randn(12, 6).astype(np.float32)), dev(np.abs(rs.randn(12, 6)).astype(np.float32))
    m2, s2 = m1.clone(), s1.clone()
    ec, idx, el = pl.update_distribution(dev(costs), dev(pool), K, m1, s1, dev(kc) if n_keep else None, dev(ka) if n_keep else None)
    all_c = dev(np.concatenate([costs, kc]))
    all_a = dev(np.concatenate([pool, ka]))
    ec2, idx2 = pl.topk_sorted(all_c, K)
    el2 = pl.gather_refit(all_a, idx2, m2, s2)
    assert torch.equal(idx, idx2.to(idx.dtype))
    assert np.array_equal(np_(ec), np_(ec2), equal_nan=True)
    assert torch.equal(el, el2) and torch.equal(m1, m2) and torch.equal(s1, s2)


def test_in_library_exchange_wait_is_bounded(monkeypatch):
    """A rank whose peer never pushes must not hang the GPU: the device-side wait gives up after its poll budget, sets
    the block's status word, and the step finishes (with garbage)."""
    import ctypes as C
    from icem_amd import IcemPlanner, _lib as L
    L.set_option("xchg_max_polls", 2000)
    pls = [_xchg_planner(r, 2, "f32") for r in range(2)]
    IcemPlanner.connect_exchange_local(pls)
    pl = pls[0]
    pl.obs0.copy_(torch.as_tensor(0.1 * np.random.RandomState(0).randn(17), dtype=pl.dt))
    st = pl._stream()
    for it in range(pl.cfg.opt_iters):  # rank 1 never runs
        L.check(pl.lib.icem_plan_iter_local(pl._h, C.byref(pl._cb), 0, it, st))
        L.check(pl.lib.icem_plan_iter_merge(pl._h, C.byref(pl._cb), 0, it, st))
    torch.cuda.synchronize()
    # the next sharded step refuses to plan on: the status word is host memory, checked without a copy or a sync
    rc = pl.lib.icem_plan_step_sharded(pl._h, C.byref(pl._cb), 1, st)
    assert rc != 0 and b"timed out" in pl.lib.icem_last_error()
    assert pl.exchange_status()[0] == 1
    assert pl.exchange_status()[0] == 0  # read clears


@pytest.mark.parametrize("h,d,o,kind,mode,N,iters", [(30, 6, 17, 0, "sum", 4096, 5), (30, 6, 17, 1, "best", 1000, 3), (30, 6, 18, 1, "sum", 8000, 3),
                                                     (12, 6, 17, 0, "final", 300, 4), (13, 4, 17, 1, "sum", 5001, 3), (30, 6, 17, 0, "sum", 17, 2),
                                                     (30, 17, 24, 1, "sum", 2000, 2),
                                                     # populations that fill the 256 slabs exactly: from the second MPC step on the
                                                     # shifted elites sit in list-less workgroups behind them (2 and 4 tiles per slab)
                                                     (30, 6, 17, 0, "sum", 8192, 2), (30, 6, 17, 1, "best", 16384, 2)])
def test_small_population_kernel_equals_two_kernel_path(h, d, o, kind, mode, N, iters, monkeypatch):
    """The small-population launch (k_iter_small.hip: a row sampled by a quad of lanes with DPP exchange of its draws,
    rollout on Tile4 = VALU + DPP row broadcast, four trajectories per wave) against the sampler + rollout16 kernels
    (one thread per row, Tile16 = six dependent f32 MFMAs per step; option fuse_max_rw = 0): the same draws, the same fmaf
    chains in the same order (an f32 MFMA is bitwise an fmaf chain over its slots) -- every buffer identical over three
    MPC steps."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env, humanoid_standup_env
    env = humanoid_standup_env(o) if d == 17 else halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind)

    def run(max_rw):
        from icem_amd import _lib as L
        L.set_option("fuse_max_rw", max_rw)
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=11, cost_mode=mode),
                         env.action_space.low[:d], env.action_space.high[:d])
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        out = []
        for s in range(3):
            a = np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(o))).copy()
            n_last = pl.population_sizes[-1]
            ea, ec = pl.current_elites()
            out.append((a, np_(pl.mean), np_(pl.std), np_(pl.costs[:n_last]), np_(pl.actions[:n_last]), np_(ea), np_(ec), np_(pl.best_cost)))
        return out

    ref, got = run(0), run(8)
    for s, (r, g) in enumerate(zip(ref, got)):
        for k, (x, y) in enumerate(zip(r, g)):
            assert np.array_equal(x, y), (s, k, np.abs(x - y).max())


def test_bench_two_ranks_without_a_launcher():
    """The multi-GPU bench command exactly as the driver types it -- `python bench.py --gpus 2`, no torch.distributed.run
    in front -- on whatever GPUs this box has (one: both ranks share it and rendezvous over gloo): bench.py starts its own
    ranks, rank 0 prints ONE line with the weak-scaling headline, the strong-scaling leg of BASELINE configs[3] and, per
    leg, which exchange ran, its latency, the block kind and the timeout count."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_population"] == 2 * 4096
    assert j["launched_by"].startswith("bench.py")
    for leg in (j, j["strong"]) + ((j["also"],) if "also" in j else ()):
        ex = leg["exchange"]
        assert ex["ran"] in ("ipc", "rccl", "host")
        if ex["ran"] == "ipc":
            assert ex["timeouts"] == 0 and ex["latency_us"] > 0 and ex["host_collectives_in_timed_loop"] == 0
    assert j["strong"]["scaling"] == "strong" and j["strong"]["n_gpus"] == 2
    if torch.cuda.device_count() >= 2:
        assert j["strong"]["global_population"] == 65536 and "also" in j


@pytest.mark.parametrize("N", [4096, 40000])
def test_noise_drawn_ahead_is_tied_to_its_stream(N):
    """The first noise of MPC step s + 1 is drawn by a launch of step s (beside the last merge at small populations, beside
    the last rollout at large ones).  A caller that enqueues step s + 1 on ANOTHER stream is not ordered behind that launch:
    the library treats the stashed noise as a miss and redraws it on the new stream -- same counters, same bits -- instead
    of reading a half-written buffer.  Steps alternate between two streams (the caller orders the streams
    for the buffers it shares between them, as it must); every result equals the single-stream run's."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6, kind=1)

    def mk():
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=N, opt_iters=3, dtype="f32", seed=11),
                         env.action_space.low, env.action_space.high)
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        return pl
    obs = [0.1 * np.random.RandomState(40 + s).randn(17) for s in range(5)]
    ref = mk()
    want = []
    for ob in obs:
        want.append((np_(ref.plan_step(ob)).copy(), np_(ref.mean).copy(), np_(ref.std).copy()))
    torch.cuda.synchronize()
    pl = mk()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    prev = torch.cuda.current_stream()
    for s, ob in enumerate(obs):
        st = streams[s & 1]
        st.wait_stream(prev)
        with torch.cuda.stream(st):
            act = pl.plan_step(ob)
        prev = st
        st.synchronize()
        assert np.array_equal(np_(act), want[s][0]), s
        assert np.array_equal(np_(pl.mean), want[s][1]) and np.array_equal(np_(pl.std), want[s][2]), s


def test_profile_overhead_calibration():
    """icem_profile_overhead: the event pair around a kernel of known duration costs more than the kernel, by a few
    microseconds that do not depend on the kernel's length (what bench.py subtracts from every per-kernel time)."""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(17)
    pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=64, opt_iters=2, dtype="f32"), env.action_space.low, env.action_space.high)
    brackets = []
    for spin in (10.0, 30.0):
        pair, kern = pl.profile_overhead(50, spin)
        assert spin <= kern < spin + 1.0, (spin, kern)
        assert kern < pair < kern + 20.0, (pair, kern)
        brackets.append(pair - kern)
    assert abs(brackets[0] - brackets[1]) < 1.5, brackets


@pytest.mark.parametrize("h,d,o,kind,mode,N,iters", [(30, 6, 17, 0, "sum", 16384, 3), (30, 6, 17, 1, "best", 40000, 4), (30, 6, 18, 1, "sum", 16384, 2),
                                                     (30, 17, 24, 1, "sum", 16384, 2)])
def test_noise_ahead_pipeline_equals_the_default_path(h, d, o, kind, mode, N, iters, monkeypatch):
    """The noise-ahead pipeline of large populations (plan.hip::plan_step_ahead, k_rollout_ahead.hip: one launch per
    iteration whose rollout workgroups run the previous merge in their prologue and map the pool's raw noise to actions as
    they load it, beside noise workgroups that draw the NEXT sampling call and, at iteration 0, a shifted-elites workgroup)
    against the sampler + rollout pair (option noise_ahead = 0): same draws, same fmaf / v_med3 per sample, same rollout code --
    every buffer identical over four MPC steps (the noise of step s + 1's first iteration is drawn during step s).  (The
    two-tile shape d = 17, o = 24 is routed to the pair whatever the switch says -- its launch spilled and lost, EXPERIMENTS
    R4.6 -- and stays here as the check that the routing leaves its results alone.)"""
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env, humanoid_standup_env
    env = humanoid_standup_env(o) if d == 17 else halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind)

    def run(on):
        from icem_amd import _lib as L
        L.set_option("noise_ahead", 1 if on else 0)
        pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=5, cost_mode=mode),
                         env.action_space.low[:d], env.action_space.high[:d])
        pl.set_model(model.kind, model.A, model.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        out = []
        for s in range(4):
            act = np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(o))).copy()
            torch.cuda.synchronize()
            n_last = pl.population_sizes[-1]
            ea, ec = pl.current_elites()
            out.append([act, np_(pl.mean), np_(pl.std), np_(ea), np_(ec), np_(pl.costs[:n_last]), np_(pl.actions[:n_last]), np_(pl.best_cost)])
        return out
    for got, want in zip(run(True), run(False)):
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
