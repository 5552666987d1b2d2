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
    # the planes carry A and B scaled by their own powers of two: a model in other units is served all the same
    assert _planner(65536, 5, scale=1e-3)[0].tile_arith == 1
    assert _planner(65536, 5, scale=40.0)[0].tile_arith == 1
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
        assert pl.tile_arith == 1
    rs = np.random.RandomState(3)
    n = 2048
    act = bound * rs.uniform(-1, 1, (n, h, d))
    obs0 = obs_scale * rs.randn(o)
    # the planner learns the bounds' magnitude from its plan buffers: one planning step first
    pl.plan_step(obs0)
    om = O.SyntheticModel(model.A, model.B, model.kind)
    oc = O.CostSpec.halfcheetah(o)
    got = np_(pl.rollout_cost(obs0, act))
    ref = O.rollout_costs(om, oc, obs0, act)
    mag = O.rollout_cost_magnitudes(om, oc, obs0, act)
    assert np.all(np.isfinite(got))
    assert np.all(np.abs(got - ref) <= RTOL * mag), float((np.abs(got - ref) / mag).max())


def test_a_state_that_leaves_fp16_range_ranks_last():
    """A trajectory whose state grows beyond 2^11 x max(|obs0|, action bound) inside the horizon overflows the planes: its
    cost comes back non-finite and its key ranks behind every finite one (icem.py:199 would rank a huge finite cost last as
    well); the other trajectories are untouched."""
    from icem_amd import IcemConfig, IcemPlanner, halfcheetah_env
    h, d, o = 30, 6, 17
    env = halfcheetah_env(o)
    A = 1.5 * np.eye(o)       # 1.5^30 = 1.9e5: every state entry that starts away from zero blows up
    A[8, 8] = 0.5             # ... but not the scored velocity of rows whose other entries start at zero
    B = np.zeros((d, o))
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=65536, opt_iters=2, dtype="f32"), env.action_space.low, env.action_space.high)
    pl.set_model(0, A, B)
    pl.set_cost_spec(env.cost_spec)
    pl.reset()
    assert pl.tile_arith == 1
    act = np.zeros((64, h, d))
    obs0 = np.zeros(o)
    obs0[8] = 1.0
    fine = np_(pl.rollout_cost(obs0, act))
    assert np.all(np.isfinite(fine))
    obs0[3] = 1.0
    blown = np_(pl.rollout_cost(obs0, act))
    assert not np.any(np.isfinite(blown)) or np.all(blown > 1e30) or np.all(np.isnan(blown))
    c = torch.as_tensor(np.concatenate([fine[:5], blown[:5]]), dtype=torch.float32, device="cuda")
    _, idx = pl.topk_sorted(c.cpu().numpy(), 10)
    assert sorted(idx.cpu().numpy()[:5].tolist()) == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("N,iters,kind,mode", [(40000, 3, 0, "sum"), (16384, 2, 1, "best"), (9000, 2, 0, "final"), (4096, 3, 1, "sum"), (700, 2, 0, "sum")])
def test_every_launch_shape_computes_the_same_bits_in_fp16_planes(N, iters, kind, mode, monkeypatch):
    """One arithmetic per handle: with the fp16 planes switched on, the noise-ahead launches, the single-launch kernel (whose
    slabs then roll out on Tile16H instead of the VALU twin of the exact tile) and the sampler + rollout pair leave the same
    bits in every buffer over three MPC steps -- at populations that take each of them by default."""
    def run(ahead, fuse):
        monkeypatch.setenv("ICEM_NOISE_AHEAD", "1" if ahead else "0")
        if fuse:
            monkeypatch.delenv("ICEM_FUSE_MAX_RW", raising=False)
        else:
            monkeypatch.setenv("ICEM_FUSE_MAX_RW", "0")
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
