"""The whole MPC step of a small population as ONE launch inside one XCD (k_step_xcd.hip; option step_xcd = 1, off by default: it
measured slower than one launch per iteration -- EXPERIMENTS R6.4 -- and stays in the tree as the record of that experiment):
the body of MpcICem.get_action (icem/controllers/icem.py:106-189) with every iteration boundary a 32-arrival barrier in one
XCD's L2.  Held here: every buffer bit for bit the default path's over several MPC steps (executed action, best cost, mean,
std, elites, last pool and costs), with shifted and kept elites, partial slabs, one iteration, both tile arithmetics; switching
between the two paths from step to step; no bounded wait ever runs out."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(N, iters, kind=0, mode="sum", arith=None, hdo=(30, 6, 17), **cfg):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    h, d, o = hdo
    env = halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind)
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=5, cost_mode=mode, **cfg),
                     env.action_space.low[:d], env.action_space.high[:d])
    pl.set_model(model.kind, model.A, model.B)
    pl.set_cost_spec(env.cost_spec)
    if arith is not None:
        pl.set_tile_arith(arith)
    pl.reset()
    return pl


def _state(pl):
    n_last = pl.population_sizes[-1]
    ea, ec = pl.current_elites()
    f = lambda t: t.detach().cpu().numpy().copy()
    return [f(pl.executed), f(pl.best_cost), f(pl.mean), f(pl.std), f(ea), f(ec), f(pl.costs[:n_last]), f(pl.actions[:n_last])]


@pytest.mark.parametrize("N,iters,kind,mode,arith,cfg", [
    (4096, 5, 0, "sum", None, {}),
    (4096, 5, 1, "best", None, {}),
    (4096, 2, 0, "sum", "f32", {}),
    (1000, 3, 0, "final", None, {}),
    (300, 4, 1, "sum", None, {}),
    (2500, 1, 0, "sum", None, {}),
    (3000, 3, 0, "sum", None, dict(keep_previous_elites=False, shift_elites=False, use_mean_actions=False)),
    (4096, 3, 0, "sum", None, dict(noise_beta=0.0)),
])
def test_one_launch_step_equals_the_launches_per_iteration(N, iters, kind, mode, arith, cfg):
    from icem_amd import _lib as L
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the one-launch step is built around 8 XCDs x 32 CUs")
    L.set_option("step_xcd", 0)
    ref = _make(N, iters, kind, mode, arith, **cfg)
    new = _make(N, iters, kind, mode, arith, **cfg)
    mixed = _make(N, iters, kind, mode, arith, **cfg)
    for s in range(5):
        obs = 0.1 * np.random.RandomState(s).randn(17)
        L.set_option("step_xcd", 0)
        ref.plan_step(obs)
        L.set_option("step_xcd", 1)
        new.plan_step(obs)
        L.set_option("step_xcd", s % 2)      # the two paths in turn: what one step draws ahead is not the other's
        mixed.plan_step(obs)
        torch.cuda.synchronize()
        for k, (x, y, z) in enumerate(zip(_state(new), _state(ref), _state(mixed))):
            assert np.array_equal(x, y, equal_nan=True), (s, k)
            assert np.array_equal(z, y, equal_nan=True), ("mixed", s, k)
    assert new.step_status() == (5, False) and ref.step_status() == (0, False) and mixed.step_status() == (2, False)


def test_what_the_one_launch_step_does_not_serve_keeps_the_default_path():
    from icem_amd import _lib as L
    L.set_option("step_xcd", 1)
    big = _make(16384, 2)                  # more rows than one XCD's 32 slabs hold
    big.plan_step(np.zeros(17))
    two_tiles = _make(1024, 2, hdo=(30, 6, 18))
    two_tiles.plan_step(np.zeros(18))
    torch.cuda.synchronize()
    assert big.step_status()[0] == 0
    assert two_tiles.step_status()[1] is False
