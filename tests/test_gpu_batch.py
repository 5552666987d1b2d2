"""``icem_plan_step_batch``: B independent planners of one configuration advanced together -- the reference's parallel episodes,
each controller its own ``get_action`` (icem/misc/rollout_utils.py:46-58, 129-152; icem/controllers/icem.py:106-189) -- with every
stage of the small-population path ONE launch for all of them (grid.y = the problem).  Held here: every problem's outputs are bit
for bit those of its own ``icem_plan_step`` (executed action, best cost, mean, std, elite set and costs, last pool and its costs)
over several MPC steps; problems differ in model, cost weights, seed, bounds and observation; what is refused is refused before
anything is launched."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _make(i, N, iters, h=30, d=6, o=17, kind=0, mode="sum", arith=None, beta=0.25):
    from icem_amd import DeviceSyntheticModel, IcemConfig, IcemPlanner, halfcheetah_env
    env = halfcheetah_env(o)
    model = DeviceSyntheticModel.make(o, d, kind=kind, seed_a=10 + i, seed_b=20 + i)
    bound = 1.0 if i % 2 == 0 else 0.5
    pl = IcemPlanner(IcemConfig(horizon=h, act_dim=d, num_traj=N, opt_iters=iters, dtype="f32", seed=100 + 7 * i, cost_mode=mode,
                                noise_beta=beta), bound * env.action_space.low[:d], bound * env.action_space.high[:d])
    pl.set_model(model.kind, model.A, model.B)
    c = env.cost_spec
    pl.set_cost(c.ctrl_weight * (1 + 0.1 * i), c.lin_idx, c.lin_weight, c.flip_idx, c.flip_penalty, c.flip_thresh)
    if arith is not None:
        pl.set_tile_arith(arith)
    pl.reset()
    return pl


def _state(pl):
    n_last = pl.population_sizes[-1]
    ea, ec = pl.current_elites()
    return [np_(pl.executed).copy(), np_(pl.best_cost).copy(), np_(pl.mean).copy(), np_(pl.std).copy(), np_(ea).copy(), np_(ec).copy(),
            np_(pl.costs[:n_last]).copy(), np_(pl.actions[:n_last]).copy()]


@pytest.mark.parametrize("B,N,iters,kind,mode,arith,hdo", [
    (8, 4096, 5, 0, "sum", None, (30, 6, 17)),      # the metric's population, eight problems
    (4, 4096, 5, 1, "best", None, (30, 6, 17)),
    (2, 4096, 3, 0, "final", "f32", (30, 6, 17)),   # the exact tile
    (3, 1000, 4, 1, "sum", None, (30, 6, 18)),      # partial slabs, o = 18
    (5, 700, 3, 0, "sum", None, (12, 6, 17)),       # PlaNet-horizon shape
    (16, 512, 2, 0, "sum", None, (13, 4, 17)),
    (2, 8192, 2, 0, "sum", None, (30, 6, 17)),
    (12, 4096, 5, 0, "sum", None, (30, 6, 17)),     # enough rows together for the noise-ahead launches (k_rollout_ahead.hip)
    (8, 4096, 3, 1, "best", "ahead", (30, 6, 17)),  # ... forced onto them below their threshold
    (2, 8192, 2, 0, "sum", "ahead", (30, 6, 17)),
])
def test_every_problem_of_a_batch_equals_its_solo_run_bit_for_bit(B, N, iters, kind, mode, arith, hdo):
    from icem_amd import IcemPlanner, _lib as L
    if arith == "ahead":
        L.set_option("batch_ahead_min_rows", 0)
        arith = None
    h, d, o = hdo
    solo = [_make(i, N, iters, h, d, o, kind, mode, arith) for i in range(B)]
    batch = [_make(i, N, iters, h, d, o, kind, mode, arith) for i in range(B)]
    for s in range(4):
        obs = [0.1 * (1 + i) * np.random.RandomState(1000 * s + i).randn(o) for i in range(B)]
        for i in range(B):
            solo[i].plan_step(obs[i])
        IcemPlanner.plan_step_batch(batch, obs)
        torch.cuda.synchronize()
        for i in range(B):
            for k, (x, y) in enumerate(zip(_state(batch[i]), _state(solo[i]))):
                assert np.array_equal(x, y, equal_nan=True), (s, i, k)
        # the problems ARE different problems
        assert not np.array_equal(np_(batch[0].executed), np_(batch[1].executed))
    # steady state: the argument blocks of a step differ from those of six steps earlier (elite buffers alternate per iteration,
    # the noise-ahead launches rotate three pools per step) only by the step's base, which travels in the kernel arguments --
    # nothing is uploaded any more
    for s in range(4, 12):
        IcemPlanner.plan_step_batch(batch, [0.1 * np.random.RandomState(s + i).randn(o) for i in range(B)])
    before = batch[0].batch_uploads
    for s in range(12, 24):
        IcemPlanner.plan_step_batch(batch, [0.1 * np.random.RandomState(s + i).randn(o) for i in range(B)])
    torch.cuda.synchronize()
    assert batch[0].batch_uploads == before, (before, batch[0].batch_uploads)
    # a planner that left a batch goes on alone, one that joins was advanced alone: same bits either way
    for i in range(B):
        solo[i].mpc_step = batch[i].mpc_step


def test_a_batch_member_can_continue_alone_and_rejoin():
    from icem_amd import IcemPlanner
    B, N, iters, o = 3, 2048, 3, 17
    ref = [_make(i, N, iters) for i in range(B)]
    mix = [_make(i, N, iters) for i in range(B)]
    rs = np.random.RandomState(5)
    for s in range(6):
        obs = [0.2 * rs.randn(o) for _ in range(B)]
        for i in range(B):
            ref[i].plan_step(obs[i])
        if s % 2 == 0:
            IcemPlanner.plan_step_batch(mix, obs)
        else:
            for i in range(B):
                mix[i].plan_step(obs[i])
        torch.cuda.synchronize()
        for i in range(B):
            for k, (x, y) in enumerate(zip(_state(mix[i]), _state(ref[i]))):
                assert np.array_equal(x, y, equal_nan=True), (s, i, k)


def test_what_a_batch_cannot_do_is_refused_before_anything_runs():
    from icem_amd import IcemPlanner, _lib as L
    a, b = _make(0, 4096, 3), _make(1, 4096, 3)
    c = _make(2, 2048, 3)                       # another population
    with pytest.raises(L.IcemError) as e:
        IcemPlanner.plan_step_batch([a, c], [np.zeros(17)] * 2)
    assert e.value.code == L.ICEM_E_INVALID
    big = [_make(i, 65536, 2) for i in range(2)]   # noise-ahead launches: they fill the chip alone
    with pytest.raises(L.IcemError) as e:
        IcemPlanner.plan_step_batch(big, [np.zeros(17)] * 2)
    assert e.value.code == L.ICEM_E_UNSUPPORTED
    for pl in big:
        assert pl.mpc_step == 0
    f64 = []
    from icem_amd import DeviceSyntheticModel, IcemConfig, halfcheetah_env
    env = halfcheetah_env(17)
    for i in range(2):
        m = DeviceSyntheticModel.make(17, 6)
        pl = IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=256, opt_iters=2, dtype="f64"), env.action_space.low, env.action_space.high)
        pl.set_model(m.kind, m.A, m.B)
        pl.set_cost_spec(env.cost_spec)
        pl.reset()
        f64.append(pl)
    with pytest.raises(L.IcemError):
        IcemPlanner.plan_step_batch(f64, [np.zeros(17)] * 2)
    with pytest.raises(L.IcemError):
        IcemPlanner.plan_step_batch([a, a], [np.zeros(17)] * 2)
    # ... and the planners are still good for a step of their own and a proper batch
    ref = [_make(0, 4096, 3), _make(1, 4096, 3)]
    obs = [0.1 * np.ones(17), -0.1 * np.ones(17)]
    for i in range(2):
        ref[i].plan_step(obs[i])
    IcemPlanner.plan_step_batch([a, b], obs)
    torch.cuda.synchronize()
    for x, y in zip(_state(a) + _state(b), _state(ref[0]) + _state(ref[1])):
        assert np.array_equal(x, y)


def test_controllers_get_action_batch_equals_their_own_get_action():
    """The host mirror of the batch: ``MpcICemHip.get_action_batch(controllers, observations)`` leaves every controller where its
    own ``get_action`` would (the reference's parallel episodes, rollout_utils.py:129-152) -- same executed actions, bit for bit."""
    from icem_amd import DeviceSyntheticModel, MpcICemHip, halfcheetah_env

    def make(i):
        env = halfcheetah_env(17)
        c = MpcICemHip(env=env, forward_model=DeviceSyntheticModel.make(17, 6, seed_a=30 + i, seed_b=40 + i), horizon=30,
                       num_simulated_trajectories=2048, factor_decrease_num=1.25, cost_along_trajectory="sum", seed=9 + i,
                       action_sampler_params=dict(alpha=0.1, elites_size=10, opt_iterations=3, init_std=0.5, use_mean_actions=True,
                                                  keep_previous_elites=True, shift_elites_over_time=True, fraction_elites_reused=0.3,
                                                  noise_beta=0.25))
        c.beginning_of_rollout(observation=np.zeros(17), state=None, mode="train")
        return c
    solo = [make(i) for i in range(4)]
    batch = [make(i) for i in range(4)]
    rs = np.random.RandomState(3)
    for s in range(4):
        obs = [0.1 * rs.randn(17) for _ in range(4)]
        want = [c.get_action(ob, None) for c, ob in zip(solo, obs)]
        got = MpcICemHip.get_action_batch(batch, obs)
        for w, g in zip(want, got):
            assert g.dtype == np.float64 and np.array_equal(w, g)
        for a, b in zip(solo, batch):
            assert np.array_equal(a.mean, b.mean) and a.last_min_cost == b.last_min_cost
