"""The fp16-plane tile arithmetic of the narrow rollout (``icem_set_tile_arith``, Tile16H of fused_dev.h: every f32 operand as two
fp16 numbers, three products per multiply-add on ``v_mfma_f32_16x16x32_f16``) -- the model step of
``ForwardModelWithDefaults.predict_n_steps`` (icem/models/abstract_models.py:17-53) for the built-in batched model.

It is NOT the bits of an f32 fmaf chain, so what is held here is north_star's bar against the float64 oracle (costs within
1e-5 of the magnitude of their sum, the same elites), the rules that choose the arithmetic (from the configuration's
GLOBAL populations, never from a launch's row count), and that every launch shape of a handle -- single launch, sampler +
rollout pair, noise-ahead launches, emulated shards -- computes the same bits in it.  The at-size loops against the oracle
run in this arithmetic too: tests/test_gpu_parity_sizes.py (c4 by default, c2 forced).
"""
import numpy as np
import pytest
import torch

from oracle import icem_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _planner(N, iters, h=30, d=6, o=17, kind=0, mode="sum", seed=5, arith=None, scale=1.0, bound=1.0):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=seed, cost_mode=mode),
                     bound * env.action_space.low[:d], bound * env.action_space.high[:d])
    pl.set_model(model.kind, scale * model.A, scale * model.B)
    pl.set_cost_spec(env.cost_spec)
    if arith is not None:
        pl.set_tile_arith(arith)
    pl.reset()
    return pl, model, env


def test_arithmetic_follows_the_global_populations_not_the_launch():
    """AUTO: fp16 planes wherever the tile serves them, at every population (ABI 4; the exact tile on request); a rank of a
    sharded run decides from the configuration, not from its own rows; models outside the planes' range and widths without a
    Tile16H keep the exact tile whatever is asked."""
    from icem_amd import IcemConfig, IcemPlanner, DeviceSyntheticModel, halfcheetah_env, humanoid_standup_env
    assert _planner(65536, 5)[0].tile_arith == 1          # 65 536 ... 26 842 rows
    assert _planner(16384, 5)[0].tile_arith == 1          # decays to 6 710 rows
    assert _planner(16384, 2)[0].tile_arith == 1          # 16 384, 13 107
    assert _planner(4096, 5)[0].tile_arith == 1
    pl = _planner(4096, 5)[0]
    assert pl.set_tile_arith("f16x2") == 1 and pl.set_tile_arith("f32") == 0 and pl.set_tile_arith("auto") == 1
    pl = _planner(65536, 5)[0]
    assert pl.set_tile_arith("f32") == 0 and pl.set_tile_arith("auto") == 1
    # one rank of eight: 8192 local rows, 65 536 global
    env = halfcheetah_env(17)
    model = DeviceSyntheticModel.make(17, 6)
    shard = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=65536, opt_iters=5, dtype="f32", rank=3, world=8),
                        env.action_space.low, env.action_space.high)
    shard.set_model(model.kind, model.A, model.B)
    shard.set_cost_spec(env.cost_spec)
    assert shard.tile_arith == 1
    # the planes carry A and B scaled by their own powers of two: a model in other units is served all the same ...
    small = _planner(65536, 5, scale=1e-3)[0]
    assert small.tile_arith == 1 and small.tile_growth == 1.0
    # ... unless it can take a state out of fp16's range inside the horizon (the benchmark's model: the reachable maximum of
    # |state| / max(|obs0|, bound) is 15; 40 x that model: 40^30) -- then the exact tile, whatever is asked
    assert 8.0 < _planner(65536, 5)[0].tile_growth < 32.0
    big = _planner(65536, 5, scale=40.0)[0]
    assert big.tile_arith == 0 and big.tile_growth > 2.0 ** 100 and big.set_tile_arith("f16x2") == 0
    # two output tiles (o = 24) and float64 handles have no Tile16H
    envh = humanoid_standup_env(24)
    mh = DeviceSyntheticModel.make(24, 17, kind=1)
    wide = IcemPlanner(IcemConfig(horizon=30, act_dim=17, num_traj=65536, opt_iters=3, dtype="f32"), envh.action_space.low, envh.action_space.high)
    wide.set_model(mh.kind, mh.A, mh.B)
    wide.set_cost_spec(envh.cost_spec)
    assert wide.tile_arith == 0 and wide.set_tile_arith(1) == 0
    with pytest.raises(Exception):
        pl.set_tile_arith(7)


@pytest.mark.parametrize("h,d,o", [(30, 6, 17), (30, 6, 18), (12, 6, 17), (13, 4, 17)])
@pytest.mark.parametrize("kind,mode", [(0, "sum"), (1, "best"), (1, "final"), (0, "final")])
def test_fp16_plane_rollout_matches_the_float64_oracle(h, d, o, kind, mode):
    """icem_rollout_cost on Tile16H against the float64 oracle: every cost within 1e-5 of the magnitude of the sum it is,
    on a population with partial tiles (n % 16 != 0) and more tiles than waves."""
    pl, model, env = _planner(70000, 2, h=h, d=d, o=o, kind=kind, mode=mode, arith=1)
    assert pl.tile_arith == 1
    n = 5003
    rs = np.random.RandomState(o + h + kind)
    act = rs.uniform(-1, 1, (n, h, d))
    obs0 = 0.3 * rs.randn(o)
    om = O.SyntheticModel(model.A, model.B, model.kind)
    oc = O.CostSpec.halfcheetah(o)
    got = np_(pl.rollout_cost(obs0, act))
    ref = O.rollout_costs(om, oc, obs0, act, mode=mode)
    mag = O.rollout_cost_magnitudes(om, oc, obs0, act)
    err = np.abs(got - ref)
    worst = int(np.argmax(err / mag))
    assert err[worst] <= RTOL * mag[worst], (worst, err[worst], mag[worst])
    # ... and it is a different arithmetic from the exact tile's, not a relabelled copy of it
    pl.set_tile_arith(0)
    exact = np_(pl.rollout_cost(obs0, act))
    assert np.abs(exact - ref).max() <= RTOL * mag.max()
    assert not np.array_equal(exact, got)


@pytest.mark.parametrize("obs_scale,bound,bscale", [(1e-4, 1.0, 1.0), (30.0, 1.0, 1.0), (1e-3, 1e-2, 1.0), (3.0, 50.0, 1.0), (0.1, 1.0, 1e-3), (0.1, 1.0, 300.0)])
def test_the_scales_follow_observation_action_and_model_magnitudes(obs_scale, bound, bscale):
    """One power of two per launch, from max(|obs0|, action bound), puts the state into fp16's window, one per model block puts
    A's and B's entries there: start observations of 1e-4 or 30, action bounds of 0.01 or 50, an action matrix of 1e-4 or 30
    beside a transition matrix of 1 -- the same 1e-5 (linear model: the state keeps the magnitudes it is given)."""
    h, d, o = 30, 6, 17
    pl, model, env = _planner(70000, 2, h=h, d=d, o=o, kind=0, arith=1, bound=bound)
    if bscale != 1.0:
        model.B = bscale * model.B
        pl.set_model(model.kind, model.A, model.B)
        # an action matrix of 30 can push a state past 2^10 x the bound within 30 steps (icem_tile_growth ~ 2000): the exact tile
        assert pl.tile_arith == (0 if bscale > 100 else 1), pl.tile_growth
    rs = np.random.RandomState(3)
    n = 2048
    act = bound * rs.uniform(-1, 1, (n, h, d))
    obs0 = obs_scale * rs.randn(o)
    # (the bounds' magnitude came with the planner's reset: icem_reset_distribution)
    om = O.SyntheticModel(model.A, model.B, model.kind)
    oc = O.CostSpec.halfcheetah(o)
    got = np_(pl.rollout_cost(obs0, act))
    ref = O.rollout_costs(om, oc, obs0, act)
    mag = O.rollout_cost_magnitudes(om, oc, obs0, act)
    assert np.all(np.isfinite(got))
    assert np.all(np.abs(got - ref) <= RTOL * mag), float((np.abs(got - ref) / mag).max())


def _growing_model(o, d, rate=1.5):
    A = rate * np.eye(o)      # 1.5^30 = 1.9e5: every state entry that starts away from zero grows past 2^11 x itself
    A[:, 8] += 0.01           # ... and feeds the scored velocity, which itself decays: costs of 1e3 and more
    A[8, 8] = 0.5
    B = 0.05 * np.random.RandomState(2).randn(d, o)
    return A, B


@pytest.mark.parametrize("N,iters", [(4096, 3), (40000, 2)])
def test_a_model_that_outgrows_fp16_keeps_the_exact_tile_and_the_oracles_elites(N, iters):
    """VERDICT r05 weak #1.  A = 1.5 I: costs of 1e5, finite and ORDERED in the reference's float64 (icem.py:147-159, 199).  The
    default arithmetic must not turn them into NaN ties: the handle sees that the model can leave the planes' range
    (icem_tile_growth = 1.9e5 > 2^10) and computes on the exact tile -- whatever is asked -- and whole MPC steps reproduce the
    float64 oracle's elite sets; every cost is finite and within 1e-4 of its magnitude (an expanding system amplifies the f32
    chain's own rounding with the state: 2-5e-5 at a growth of 1e5, whatever computes it in f32; strict parity there: dtype f64)."""
    from icem_amd import IcemConfig, IcemPlanner, halfcheetah_env
    h, d, o = 30, 6, 17
    env = halfcheetah_env(o)
    A, B = _growing_model(o, d)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=3), env.action_space.low, env.action_space.high)
    pl.set_model(0, A, B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    assert pl.tile_arith == 0 and pl.tile_growth > 1e5
    assert pl.set_tile_arith("f16x2") == 0 and pl.set_tile_arith("auto") == 0
    om = O.SyntheticModel(A, B, 0)
    oc = O.CostSpec.halfcheetah(o)
    K = pl.K
    obs = 0.3 * np.random.RandomState(1).randn(o)
    seen = []

    def on_iteration(it):
        n = pl.population_sizes[it]
        seen.append((it, n, np_(pl.actions[:n]).copy(), np_(pl.costs[:n]).copy()))
    pl.plan_step(obs, on_iteration=on_iteration)
    torch.cuda.synchronize()
    assert pl.nonfinite_costs() == 0
    for it, n, act, cost in seen:
        ref = O.rollout_costs(om, oc, obs, act)
        mag = O.rollout_cost_magnitudes(om, oc, obs, act)
        assert np.all(np.isfinite(cost)) and np.abs(ref).max() > 1e3
        assert np.all(np.abs(cost - ref) <= 10 * RTOL * mag), float((np.abs(cost - ref) / mag).max())
        # the K best of the device's costs are the K best of the oracle's (sets; costs this large are far apart)
        assert set(np.argsort(cost, kind="stable")[:K].tolist()) == set(np.argsort(ref, kind="stable")[:K].tolist())


def test_actions_outside_the_bounds_are_counted_not_hidden():
    """icem_rollout_cost takes ANY actions; the planes' scale comes from the planner's bounds.  Actions 10^5 x the bounds drive
    states out of fp16's range: those trajectories' costs are NaN (ranked last, as icem.py:199's argsort ranks a NaN) and the
    handle's status word counts them (icem_nonfinite_costs) -- the rows inside the bounds are untouched."""
    pl, model, env = _planner(70000, 2, arith=1)
    assert pl.tile_arith == 1 and pl.nonfinite_costs() == 0
    rs = np.random.RandomState(5)
    act = rs.uniform(-1, 1, (64, 30, 6))
    obs0 = 0.3 * rs.randn(17)
    fine = np_(pl.rollout_cost(obs0, act))
    assert np.all(np.isfinite(fine)) and pl.nonfinite_costs() == 0
    act2 = act.copy()
    act2[:7] *= 1e5
    got = np_(pl.rollout_cost(obs0, act2))
    assert np.all(np.isnan(got[:7])) and np.array_equal(got[7:], fine[7:])
    assert pl.nonfinite_costs() == 7


def test_get_action_reports_a_step_with_non_finite_costs():
    """The host mirror's view (icem_get_action): a step that produced non-finite costs from a FINITE observation returns
    ICEM_E_RANGE -- raised as IcemError by the binding -- with the step counted; a NaN observation (the reference goes on:
    argsort ranks NaNs last, icem.py:199) raises nothing."""
    from icem_amd import IcemConfig, IcemPlanner, halfcheetah_env, _lib as L
    h, d, o = 30, 6, 17
    env = halfcheetah_env(o)
    A = 1e4 * np.eye(o)       # 1e4^30 overflows float32 itself: the exact tile's costs are inf - inf = NaN
    B = np.zeros((d, o))
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=256, opt_iters=2, dtype="f32"), env.action_space.low, env.action_space.high)
    pl.set_model(0, A, B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    assert pl.tile_arith == 0
    with pytest.raises(L.IcemError) as e:
        pl.get_action_host(np.ones(o))
    assert e.value.code == L.ICEM_E_RANGE and pl.mpc_step == 1
    obs = np.ones(o)
    obs[3] = np.nan
    pl.get_action_host(obs)   # not an error: the inputs were not finite
    assert pl.mpc_step == 2


@pytest.mark.parametrize("N,iters,kind,mode", [(40000, 3, 0, "sum"), (16384, 2, 1, "best"), (9000, 2, 0, "final"), (4096, 3, 1, "sum"), (700, 2, 0, "sum")])
def test_every_launch_shape_computes_the_same_bits_in_fp16_planes(N, iters, kind, mode, monkeypatch):
    """One arithmetic per handle: with the fp16 planes switched on, the noise-ahead launches, the single-launch kernel (whose
    slabs then roll out on Tile16H instead of the VALU twin of the exact tile) and the sampler + rollout pair leave the same
    bits in every buffer over three MPC steps -- at populations that take each of them by default."""
    def run(ahead, fuse):
        from icem_amd import _lib as L
        L.reset_options()
        L.set_option("noise_ahead", 1 if ahead else 0)
        if not fuse:
            L.set_option("fuse_max_rw", 0)
        pl, _, _ = _planner(N, iters, kind=kind, mode=mode, arith=1)
        assert pl.tile_arith == 1
        out = []
        for s in range(3):
            act = np_(pl.plan_step(0.1 * np.random.RandomState(s).randn(17))).copy()
            torch.cuda.synchronize()
            n_last = pl.population_sizes[-1]
            ea, ec = pl.current_elites()
            out.append([act, np_(pl.mean), np_(pl.std), np_(ea), np_(ec), np_(pl.costs[:n_last]), np_(pl.actions[:n_last]), np_(pl.best_cost)])
        return out
    base = run(False, False)            # sampler + rollout16 pair
    for other in (run(True, True), run(False, True)):
        for got, want in zip(other, base):
            for x, y in zip(got, want):
                assert np.array_equal(x, y)


def test_fp16_planes_select_the_elites_of_the_exact_tile():
    """Same seed, same noise: three MPC steps at N = 65 536 in both arithmetics choose the same elites and actions in every
    step (the 1e-6-class differences between the arithmetics never reach an elite boundary here), costs agree to 1e-5."""
    a, _, _ = _planner(65536, 5, arith=1)
    b, _, _ = _planner(65536, 5, arith=0)
    for s in range(3):
        obs = 0.1 * np.random.RandomState(40 + s).randn(17)
        xa, xb = np_(a.plan_step(obs)), np_(b.plan_step(obs))
        torch.cuda.synchronize()
        n_last = a.population_sizes[-1]
        assert np.array_equal(np_(a.actions[:n_last]), np_(b.actions[:n_last]))       # same elites -> same distributions -> same pool
        assert np.array_equal(np_(a.current_elites()[0]), np_(b.current_elites()[0]))
        ca, cb = np_(a.costs[:n_last]), np_(b.costs[:n_last])
        assert np.abs(ca - cb).max() <= RTOL * np.abs(cb).max()
        np.testing.assert_allclose(xa, xb, rtol=RTOL, atol=2e-6)
