"""The reference's other shipped settings -- Door (d = 28, o = 39; icem/environments/mjenvs.py:57-78), Relocate (d = 30, o = 39;
mjenvs.py:155-174), FetchPickAndPlace (d = 4, o = 28; icem/environments/robotics.py:150-164), h = 30 -- on the TileHN kernel
(k_rollout_hn.hip): the batched model step of abstract_models.py:17-53 on the 16-bit matrix cores (fp16 planes), the env's
cost (icem_cost_terms: norms of slices, gated norms, hinge offset, opening / closeness bonuses) evaluated across the four
lanes of a trajectory.  Held to the float64 oracle; the whole loop at the headline population is in
tests/test_gpu_parity_sizes.py::test_full_loop_on_the_shipped_door_relocate_fpp_shapes.
"""
import numpy as np
import pytest
import torch

from oracle import icem_oracle as O

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _env(name):
    from icem_amd import envs as E
    return {"door": (E.door_env, O.CostSpec.door), "relocate": (E.relocate_env, O.CostSpec.relocate),
            "fpp": (E.fetch_pick_and_place_env, O.CostSpec.fetch_pick_and_place)}[name]


def _planner(env, model, N=4096, iters=3, mode="sum", arith=None, h=30):
    from icem_amd import IcemConfig, IcemPlanner
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=env.action_space.shape[0], num_traj=N, opt_iters=iters, dtype="f32", seed=7, cost_mode=mode),
                     env.action_space.low, env.action_space.high)
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    if arith is not None:
        pl.set_tile_arith(arith)
    pl.reset()
    return pl


@pytest.mark.parametrize("name", ["door", "relocate", "fpp"])
@pytest.mark.parametrize("kind,mode", [(0, "sum"), (1, "sum"), (1, "best"), (0, "final")])
def test_rollout_on_tilehn_matches_the_float64_oracle(name, kind, mode):
    """icem_rollout_cost: every cost within 1e-5 of the magnitude of the sum it is -- but for the few trajectories where an f32
    state lands on the other side of an indicator term's threshold than the float64 one (a whole bonus apart: counted, < 1 %)."""
    from icem_amd import DeviceSyntheticModel
    mk, spec_fn = _env(name)
    env, spec = mk(), spec_fn()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    pl = _planner(env, model, mode=mode)
    assert pl.tile_arith == 1           # the TileHN kernel serves it
    om = O.SyntheticModel(model.A, model.B, model.kind)
    rs = np.random.RandomState(3 + kind)
    obs0 = 0.2 * rs.randn(o)
    acts = rs.uniform(-1, 1, (16 * 33 + 5, 30, d)) * env.action_space.high     # partial last tile, more tiles than one wave takes
    got = np_(pl.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)))
    want = O.rollout_costs(om, spec, obs0, acts, mode=mode).astype(np.float64)
    mag = O.rollout_cost_magnitudes(om, spec, obs0, acts)
    ok = np.abs(got - want) <= 1e-5 * mag
    assert ok.mean() > 0.99, (ok.mean(), np.abs(got - want).max())
    if mode == "sum":
        assert np.median(np.abs(got - want) / mag) < 2e-6
    # the exact-f32 GEMM kernel on request (icem_set_tile_arith 0): the same costs to f32 rounding
    assert pl.set_tile_arith("f32") == 0
    exact = np_(pl.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)))
    close = np.abs(exact - got) <= 1e-4 * (1 + np.abs(want))
    assert close.mean() > 0.99 and not np.array_equal(exact, got)


@pytest.mark.parametrize("name", ["door", "relocate", "fpp"])
def test_whole_mpc_steps_on_tilehn(name):
    """icem_plan_step (sampler with the merge in its prologue + TileHN rollout + last merge): the last pool of every step --
    sampled rows and shifted-elite rows -- re-scored by the oracle, and the launches of icem_plan_step against the split API."""
    from icem_amd import DeviceSyntheticModel
    mk, spec_fn = _env(name)
    env, spec = mk(), spec_fn()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=1)
    om = O.SyntheticModel(model.A, model.B, model.kind)
    pl, ref = _planner(env, model, N=2000, iters=3), _planner(env, model, N=2000, iters=3)
    for s_ in range(3):
        ob = 0.2 * np.random.RandomState(20 + s_).randn(o)
        a0 = np_(pl.plan_step(ob))
        a1 = np_(ref.plan_step(ob, on_iteration=lambda it: None))   # the split API: one C call per stage
        assert np.all(np.isfinite(a0)) and np.array_equal(a0, a1)
        n_last = pl.population_sizes[-1]
        assert np.array_equal(np_(pl.costs[:n_last]), np_(ref.costs[:n_last]))
        pool = np_(pl.actions[:n_last])
        rescored = O.rollout_costs(om, spec, ob.astype(np.float32).astype(np.float64), pool).astype(np.float64)
        mag = O.rollout_cost_magnitudes(om, spec, ob.astype(np.float32).astype(np.float64), pool)
        dev = np_(pl.costs[:n_last])
        assert (np.abs(dev - rescored) <= 1e-5 * mag).mean() > 0.99


def test_term_lists_outside_the_compiled_programs_keep_the_gemm_kernel():
    """A term list that fits none of TileHN's compiled programs (two slices longer than four entries), a health or difference
    term, or another horizon: the handle stays on the exact-f32 GEMM kernel, silently, with the same costs."""
    from icem_amd import DeviceSyntheticModel, envs as E
    env = E.door_env()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=0)
    assert _planner(env, model).tile_arith == 1
    assert _planner(env, model, h=12).tile_arith == 0
    import dataclasses
    two_long = dataclasses.replace(env.cost_spec, terms=env.cost_spec.terms + (E.CostTerm(E.TERM_SUMSQ, 0, -1, 20, 1e-3),))
    env2 = E.SyntheticEnv("Door+", o, env.action_space.low, env.action_space.high, two_long)
    pl2 = _planner(env2, model)
    assert pl2.tile_arith == 0
    rs = np.random.RandomState(1)
    obs0, acts = 0.2 * rs.randn(o), rs.uniform(-1, 1, (100, 30, d))
    got = np_(pl2.rollout_cost(obs0, torch.as_tensor(acts, dtype=pl2.dt, device=pl2.device)))
    spec2 = dataclasses.replace(O.CostSpec.door(), terms=O.CostSpec.door().terms + (O.CostTerm(O.TERM_SUMSQ, 0, -1, 20, 1e-3),))
    want = O.rollout_costs(O.SyntheticModel(model.A, model.B, model.kind), spec2, obs0, acts).astype(np.float64)
    assert (np.abs(got - want) <= 1e-4 * (1 + np.abs(want))).mean() > 0.98


@pytest.mark.parametrize("name,kind,mode", [("door", 1, "sum"), ("relocate", 1, "best"), ("fpp", 0, "final"), ("fpp", 1, "sum")])
def test_every_wave_arrangement_of_a_tile_computes_the_same_bits(name, kind, mode, monkeypatch):
    """The three arrangements of TileHN's work -- rollout_hn_split_kernel (at most one tile per CU: a model wave per OUTPUT TILE
    exchanging operand planes through LDS + a cost wave), rollout_hn_pair_kernel (at most two tiles per CU: a model wave and a
    cost wave; ICEM_HN_SPLIT=0) and rollout_hn_kernel (one wave per tile; ICEM_HN_PAIR=0) -- run the same operations on the same
    values: every cost, every elite, mean, std and executed action bit for bit over three MPC steps, at N = 4096 (256 + 1
    tiles from the second step on: one workgroup walks two) and N = 6000 (the pair form at two tiles per workgroup), and for a
    stand-alone rollout with a ragged last tile."""
    from icem_amd import DeviceSyntheticModel
    mk, _ = _env(name)
    env = mk()
    o, d = env.obs_dim, env.action_space.shape[0]
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    rs = np.random.RandomState(11)
    acts = rs.uniform(-1, 1, (16 * 9 + 5, 30, d)) * env.action_space.high
    obs_r = 0.2 * rs.randn(o)
    from icem_amd import _lib as L
    variants = {"split": {}, "pair": {"hn_split": 0}, "single": {"hn_pair": 0}}
    for N in (4096, 6000):
        out = {}
        for label, envs in variants.items():
            L.reset_options()
            for k, v in envs.items():
                L.set_option(k, v)
            pl = _planner(env, model, N=N, iters=3, mode=mode)
            res = []
            for s in range(3):
                obs = 0.2 * np.random.RandomState(50 + s).randn(o)
                a = pl.plan_step(obs).cpu().numpy().copy()
                ea, ec = pl.current_elites()
                res.append((a, pl.costs.cpu().numpy().copy(), ea.cpu().numpy().copy(), ec.cpu().numpy().copy(),
                            pl.mean.cpu().numpy().copy(), pl.std.cpu().numpy().copy()))
            res.append((pl.rollout_cost(obs_r, torch.as_tensor(acts, dtype=pl.dt, device=pl.device)).cpu().numpy().copy(),))
            out[label] = res
        for label in ("split", "pair"):
            for s, (x, y) in enumerate(zip(out[label], out["single"])):
                for k, (u, v) in enumerate(zip(x, y)):
                    assert np.array_equal(u, v, equal_nan=True), (label, N, s, k)
