"""Host-side mirror of the reference's controller interface for the hot path.

``MpcICemHip`` keeps the constructor kwargs, ``beginning_of_rollout`` / ``get_action`` /
``end_of_rollout`` signatures, attributes (``has_state``, ``needs_data`` ...) and error
behaviour of the reference's ``MpcICem`` (icem/controllers/icem.py:15-247 on top of
icem/controllers/mpc.py:21-83 and icem/controllers/abstract_controller.py:43-91), so the
episode runner (icem/misc/rollout_utils.py:166-216) and ``main.py`` drive it unchanged.  The
arithmetic inside the CEM loop runs in ``libicem_hip.so``; nothing here computes on the CPU
except the optional *foreign* forward model (a reference-style CPU simulator), which is the
caller's code.

Two execution paths, chosen by the forward model:

* device path  -- ``forward_model`` is a :class:`~icem_amd.models.DeviceSyntheticModel` and the
  env carries a ``cost_spec``: the whole MPC step (all CEM iterations) is enqueued on one HIP
  stream with no host synchronisation; only ``obs`` goes in and the action comes out.
* torch-model path -- a device-resident ``torch.nn.Module`` model (``TorchForwardModel``): sampling,
  cost reduction, top-k and refit in HIP, the learned model's batched steps in torch on the same GPU.
* host-model path -- any object with the reference's ``predict_n_steps`` contract: sampling,
  top-k and refit run on the GPU, the model/cost run wherever the model runs.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Mapping
from importlib import import_module
from typing import Callable, Optional, Union
from warnings import warn

import numpy as np
import torch

from .envs import Box, Discrete
from .models import DeviceSyntheticModel, TrajectoryBatch
from .planner import IcemConfig, IcemPlanner
from ._lib import COST_MODES as L_COST_MODES

try:  # the reference logs through `allogger` (icem.py:28,177); optional here
    import allogger as _allogger
except Exception:  # pragma: no cover - allogger is not in the image
    _allogger = None


class _NullLogger:
    logdir = None

    def log(self, value, key=None):
        pass

    def info(self, *a, **k):
        pass


def _get_logger(scope):
    if _allogger is not None:
        try:
            return _allogger.get_logger(scope=scope, default_outputs=["tensorboard"])
        except Exception:
            pass
    return _NullLogger()


# ---------------------------------------------------------------------------------------------
# base types (icem/misc/base_types.py:39-59, icem/controllers/abstract_controller.py)
# ---------------------------------------------------------------------------------------------

class Controller(ABC):
    needs_training = False
    needs_data = False
    has_state = False
    required_settings = []

    def __init__(self, *, env):
        self.env = env

    @abstractmethod
    def get_action(self, obs, state, mode="train"):
        """obs: observation from the environment; state: env-internal state; mode: train/eval/expert"""


class StatefulController(Controller, ABC):
    has_state = True

    @abstractmethod
    def beginning_of_rollout(self, *, observation, state=None, mode):
        pass

    @abstractmethod
    def end_of_rollout(self, total_time, total_return, mode):
        pass


class ParallelController(Controller, ABC):
    @abstractmethod
    def get_parallel_policy_copy(self, indices):
        pass


class ModelBasedController(Controller, ABC):
    def __init__(self, *, forward_model, env, cost_along_trajectory, do_visualize_plan=None,
                 use_env_reward_as_cost=False, **kwargs):
        super().__init__(env=env, **kwargs)
        self.forward_model = forward_model
        self.do_visualize_plan = do_visualize_plan
        self.cost_fn = self.env.cost_fn
        self.cost_along_trajectory = cost_along_trajectory
        self.use_env_reward_as_cost = use_env_reward_as_cost

    def visualize_plan(self, *, obs, state, acts):
        """abstract_controller.py:93-128: render a plan in a copy of the environment.  Only ground-truth environments
        with live rendering can (none exists without MuJoCo); for every other environment the reference does nothing,
        and so does this.  ``obs [h,o]`` / ``acts [h,d]``: the best trajectory of the last CEM iteration."""
        if self.do_visualize_plan is None or not self.do_visualize_plan:
            return
        env = self.env
        if not getattr(env, "supports_live_rendering", False):
            return
        viz = getattr(self, "visualize_env", None)
        if viz is None:
            # abstract_controller.py:99-107: a second instance of the env by NAME, with rendering switched on where the env
            # only enables it at construction; envs that bring their own copy constructor are asked for that instead
            from copy import deepcopy
            init_kwargs = deepcopy(getattr(env, "init_kwargs", {}) or {})
            if getattr(env, "enable_rendering_at_init", False):
                init_kwargs["enable_rendering"] = True
            factory = getattr(self, "env_from_string", None) or globals().get("env_from_string")
            if factory is not None and getattr(env, "name", None) is not None:
                viz = factory(env.name, **init_kwargs)
            else:
                viz = env.make_visualization_copy()
            self.visualize_env = viz
            viz.reset()
        if self.do_visualize_plan == "last":
            viz.set_state_from_observation(obs[-1])
            viz.step(acts[-1])
            viz.render()
        elif self.do_visualize_plan == "all":
            viz.set_GT_state(state)
            for a in acts:
                viz.step(a)
                viz.render()
        else:
            raise AttributeError("unknown mode for do_visualize_plan: Options: None, 'last','all'")

    def trajectory_cost_fn(self, cost_fn, rollout_buffer):
        """Per-trajectory cost of a batch of rollouts (abstract_controller.py:74-91)."""
        if self.use_env_reward_as_cost:
            costs_path = -np.asarray(rollout_buffer.as_array("rewards"))
            if costs_path.ndim == 3 and costs_path.shape[-1] == 1:   # predict() reports rewards as [N, 1] per step
                costs_path = costs_path[..., 0]
        else:
            costs_path = np.asarray([cost_fn(r["observations"], r["actions"], r["next_observations"])
                                     for r in rollout_buffer])
        if self.cost_along_trajectory == "sum":
            return np.sum(costs_path, axis=1)
        if self.cost_along_trajectory == "best":
            return np.amin(costs_path, axis=1)
        if self.cost_along_trajectory == "final":
            return costs_path[:, -1]
        raise NotImplementedError(
            "Implement method {} to compute cost along trajectory".format(self.cost_along_trajectory))


class _RowwiseCursor:
    """Walks ``sequences [p,h,...]`` the way ``ArrayIteratorParallelRowwise`` does (controllers/utils.py:18-51), with
    a (first row, time) cursor instead of re-slicing the array at every block:

    * ``k == 1``: one trajectory at a time -- ``[row, t]`` for t = 0..h-1, then the next row.  This is how
      ``GroundTruthModel.predict_n_steps`` consumes a policy (gt_model.py:84-90: 1-D observation, h calls per start state).
    * ``1 < k < rows left``: ``k`` rows at a time -- ``[row:row+k, t]`` for t = 0..h-1, then the next ``k`` rows.
    * ``k == rows left`` (also the last block of the case above): one column of all of them per call; when the columns
      are used up the next call raises.

    Error behaviour as upstream: more parallel rows than sequences -> AttributeError at construction; nothing left ->
    AttributeError (column-wise walk) or IndexError (``k == 1`` walk past the last row, upstream's ``array[0, t]`` on an
    empty array); a remainder smaller than ``k`` -> AssertionError.  Returned arrays are views of ``sequences``."""

    def __init__(self, sequences, k: int):
        if k > sequences.shape[0]:
            raise AttributeError("too many parallel rows requested!")
        self.sequences, self.k = sequences, k
        self.row = self.t = 0
        self.columns_used_up = sequences.shape[1] == 0

    def next(self):
        s, k = self.sequences, self.k
        if self.columns_used_up:
            raise AttributeError("I don't have any item(s) left.")
        left = s.shape[0] - self.row
        if k == 1 or k < left:
            if left <= 0:
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            out = s[self.row, self.t] if k == 1 else s[self.row:self.row + k, self.t]
            self.t += 1
            if self.t >= s.shape[1]:        # this block of rows is done
                self.t = 0
                self.row += k
            return out
        assert k == left                    # fully parallel from here on
        out = s[self.row:, self.t]
        self.t += 1
        self.columns_used_up = self.t >= s.shape[1]
        return out


class OpenLoopPolicy(ParallelController):
    """Replays ``action_sequences [p,h,d]`` (abstract_controller.py:153-184).  The number of rows served per call is
    fixed by the FIRST observation it is asked with -- 1-D: one trajectory at a time (the reference's
    ``GroundTruthModel``), ``[k,o]``: k at a time, ``[p,o]``: one column per call (batched models) -- see
    ``_RowwiseCursor``."""

    def __init__(self, action_sequences, *, env=None):
        super().__init__(env=env)
        self.action_sequences = action_sequences
        self.action_sequence_iterator = None

    @staticmethod
    def get_num_parallel(obs):
        return 1 if np.ndim(obs) == 1 else np.shape(obs)[0]

    def get_action(self, obs, state=None, mode="train"):
        if self.action_sequence_iterator is None:
            self.action_sequence_iterator = _RowwiseCursor(self.action_sequences, self.get_num_parallel(obs))
        return self.action_sequence_iterator.next()

    def get_parallel_policy_copy(self, indices):
        return OpenLoopPolicy(self.action_sequences[indices], env=self.env)


class MpcController(ModelBasedController, StatefulController, ABC):
    def __init__(self, *, horizon, num_simulated_trajectories, factor_decrease_num=1, verbose=False, **kwargs):
        super().__init__(**kwargs)
        self.horizon = horizon
        self.num_sim_traj = num_simulated_trajectories
        self.factor_decrease_num = factor_decrease_num
        if num_simulated_trajectories < 2:
            raise ValueError("At least two trajectories needed!")
        self.verbose = verbose
        self.forward_model_state = None

    def check_model_consistency(self):
        """mpc.py:39-47: a ground-truth model's state against the actual environment's (verbose mode).  Ground-truth models
        and envs are recognised by their interface (``get_GT_state`` / ``compute_state_difference``: the simulators
        themselves are out of scope); for every other pair this is a no-op, as upstream."""
        env, fm = self.env, self.forward_model
        if not (callable(getattr(env, "get_GT_state", None)) and callable(getattr(env, "compute_state_difference", None))
                and callable(getattr(fm, "set_state", None))):
            return
        model_state = self.forward_model_state
        env_state = env.get_GT_state()
        diff = env.compute_state_difference(model_state, env_state)
        if diff > 1e-5:
            print(f"Warning: internal GT model and actual env are not in sync: Difference: {diff}")
            print("env state:", env_state)
            print("model_state:", model_state)

    def simulate_trajectories(self, *, obs, state, action_sequences):
        """mpc.py:56-67: tile the start observation, wrap the actions in an open-loop policy and
        hand both to the model's ``predict_n_steps``."""
        p = action_sequences.shape[0]
        start_obs = np.array([obs] * p)
        start_states = [state] * p
        return self.forward_model.predict_n_steps(start_observations=start_obs, start_states=start_states,
                                                  policy=OpenLoopPolicy(action_sequences), horizon=self.horizon)[0]

    def beginning_of_rollout(self, *, observation, state=None, mode):
        # mpc.py:69-73: a ground-truth model starts from the env state the harness hands over
        fm = self.forward_model
        if state is not None and callable(getattr(fm, "set_state", None)) and callable(getattr(fm, "get_state", None)):
            self.forward_model_state = state
        else:
            self.forward_model_state = fm.reset(observation)

    def end_of_rollout(self, total_time, total_return, mode):
        pass

    # -- compute_new_mean (icem.py:168-171, 191-192; mpc.py:241-243, 265-269) ---------------------------------------
    def _new_mean_is_overridden(self, base) -> bool:
        return getattr(type(self), "compute_new_mean", None) is not getattr(base, "compute_new_mean", None)

    def _last_predicted_observation(self, obs, best_actions: np.ndarray) -> np.ndarray:
        """``simulated_paths[best_traj_idx]["observations"][-1]`` (icem.py:170, mpc.py:242): the observation in front of the
        best trajectory's last action.  The N rollouts of the step are never materialised: this one row is re-rolled on
        demand -- only when a subclass overrides ``compute_new_mean``."""
        acts = np.asarray(best_actions, dtype=np.float64)[None]
        if self.device_path:
            p = self.planner
            _, o = p.rollout_cost(np.asarray(obs, dtype=np.float64), torch.as_tensor(acts, dtype=p.dt, device=p.device),
                                  return_observations=True)
            return o.detach().cpu().numpy().astype(np.float64)[0, -1]
        # (two copies of the row: OpenLoopPolicy hands a single trajectory out row-wise, as the reference's does -- a host
        #  model's batched predict wants the batch form)
        batch = self.simulate_trajectories(obs=obs, state=self.forward_model_state, action_sequences=np.repeat(acts, 2, axis=0))
        return np.asarray(batch.as_array("observations"), dtype=np.float64)[0, -1]

    def _bind_models(self, world=1):
        """Which rollout path this controller's model allows: built-in device model + parametric cost (HIP rollout),
        a device torch model, or any host model through the reference's ``predict_n_steps`` interface."""
        # use_env_reward_as_cost (abstract_controller.py:76-77: costs = -rewards of the rollouts): on the device when the
        # model reports this very environment's reward, which is -cost_fn = -cost_spec -- the same kernels, the same
        # numbers; any other reward source goes through the host-model path
        reward_is_cost_spec = getattr(self.forward_model, "env", None) is self.env and hasattr(self.env, "reward_fn")
        self.device_path = (isinstance(self.forward_model, DeviceSyntheticModel)
                            and getattr(self.env, "cost_spec", None) is not None
                            and (not self.use_env_reward_as_cost or reward_is_cost_spec))
        self.rssm_path = hasattr(self.forward_model, "rollout_cost") and hasattr(self.forward_model, "params")
        self.torch_path = (not self.device_path and not self.rssm_path and hasattr(self.forward_model, "torch_step")
                           and hasattr(self.forward_model, "torch_cost"))
        if self.device_path:
            m, c = self.forward_model, self.env.cost_spec
            self.planner.set_model(m.kind, m.A, m.B)
            self.planner.set_cost_spec(c)
        elif world != 1:
            raise NotImplementedError("sharding over GPUs needs the device path (built-in model + cost_spec)")
        # a torch model without its own cost callable is scored by the env's parametric cost on the device
        self._torch_rollouts = {}
        self.torch_spec_cost = (self.torch_path and getattr(self.forward_model, "cost", None) is None
                                and getattr(self.env, "cost_spec", None) is not None)
        if self.torch_spec_cost:
            self.planner.set_cost_spec(self.env.cost_spec)

    def _costs_of(self, obs, actions: torch.Tensor) -> torch.Tensor:
        """Per-trajectory costs (device tensor) of a batch of device action sequences, through whichever
        model this controller was given."""
        p = self.planner
        if self.device_path:
            return p.rollout_cost(np.asarray(obs, dtype=np.float64), actions)
        if self.rssm_path:   # learned dynamics fused into one launch (icem_rssm_rollout_cost)
            return self.forward_model.rollout_cost(obs, actions, L_COST_MODES[p.cfg.cost_mode]).to(p.dt)
        if self.torch_path:
            # device-resident torch model (learned dynamics): h batched steps on the GPU, scored by the HIP cost
            # kernels; nothing leaves the device.  The h * (model + cost) torch launches of one population size are
            # captured in a HIP graph on first use and replayed (static input / output tensors per size).
            m = self.forward_model
            n = actions.shape[0]
            o0 = torch.as_tensor(np.asarray(obs, dtype=np.float64), dtype=m.dtype, device=p.device)
            r = self._torch_rollouts.get(n)
            if r is None:
                r = self._torch_rollouts[n] = _TorchRollout(m, p, n, self.torch_spec_cost)
            r.run(o0, actions)
            if self.torch_spec_cost:
                # the rollout stays in HBM step-major ([h+1, n, o]); icem_trajectory_cost scores it in one launch
                return p.trajectory_cost(r.buf[:p.h].transpose(0, 1), actions, r.buf[1:].transpose(0, 1))
            return p.cost_reduce(r.step_costs)
        batch = self.simulate_trajectories(obs=obs, state=self.forward_model_state,
                                           action_sequences=actions.cpu().numpy().astype(np.float64))
        return torch.as_tensor(self.trajectory_cost_fn(self.cost_fn, batch), dtype=p.dt, device=p.device)



class _TorchRollout:
    """The h batched steps of a torch dynamics model for ONE population size, with static tensors so that the whole
    chain of torch launches can be replayed as a HIP graph (``forward_model.use_graph``; a model whose step cannot be
    captured -- data-dependent control flow, host syncs -- stays on eager launches)."""

    def __init__(self, model, planner, n, spec_cost):
        self.m, self.p, self.spec_cost = model, planner, spec_cost
        dev = planner.device
        self.o0 = torch.empty((n, model.obs_dim), dtype=model.dtype, device=dev)
        self.actions = torch.empty((n, planner.h, planner.d), dtype=model.dtype, device=dev)
        if spec_cost:
            self.buf = torch.empty((planner.h + 1, n, model.obs_dim), dtype=planner.dt, device=dev)
        else:
            self.step_costs = torch.empty((n, planner.h), dtype=planner.dt, device=dev)
        self.graph = None
        self.use_graph = bool(getattr(model, "use_graph", True))

    def _chain(self):
        m, p, o = self.m, self.p, self.o0
        if self.spec_cost:
            self.buf[0] = o
        for t in range(p.h):
            a = self.actions[:, t]
            if not self.spec_cost:
                self.step_costs[:, t] = m.torch_cost(o, a).to(p.dt)
            o = m.torch_step(o, a)
            if self.spec_cost:
                self.buf[t + 1] = o

    def run(self, o0, actions):
        self.o0.copy_(o0.expand_as(self.o0))
        self.actions.copy_(actions)
        if not self.use_graph:
            return self._chain()
        if self.graph is None:
            try:
                side = torch.cuda.Stream(device=self.p.device)
                side.wait_stream(torch.cuda.current_stream(self.p.device))
                with torch.cuda.stream(side):   # warm-up off the capture (lazy initialisations, workspace allocations)
                    self._chain()
                    self._chain()
                torch.cuda.current_stream(self.p.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._chain()
                self.graph = g
            except Exception:  # noqa: BLE001 -- not capturable: eager launches from here on
                self.use_graph = False
                torch.cuda.synchronize(self.p.device)
                return self._chain()
        self.graph.replay()


# ---------------------------------------------------------------------------------------------
# the drop-in controller
# ---------------------------------------------------------------------------------------------

class MpcICemHip(MpcController):
    """iCEM with the inner loop on MI355X (drop-in for ``controllers.icem.MpcICem``).

    Extra keyword arguments (all optional, so the reference's settings JSON works untouched):
    ``dtype`` ("f32" | "f64"), ``seed``, ``rng_rounds`` (10 | 7), ``device``,
    ``noise_source`` ("philox": device counter RNG; "numpy_legacy": the reference's draws from
    the global ``np.random`` stream, in the reference's order -- parity mode; or a callable
    ``noise(num) -> (z_r, z_i)``), ``process_group`` / ``rank`` / ``world`` to shard N over GPUs.
    """

    def __init__(self, *, action_sampler_params, dtype="f32", seed=0, rng_rounds=10, device="cuda:0",
                 noise_source: Union[str, Callable] = "philox", process_group=None, rank=0, world=1,
                 deterministic_replay=False, **kwargs):
        super().__init__(**kwargs)
        self._parse_action_sampler_params(**dict(action_sampler_params))
        self._check_validity_parameters()
        self.logger = _get_logger(self.__class__.__name__)
        self.was_reset = False
        self.noise_source = noise_source
        # device noise: every episode gets its own streams unless the caller opts into replaying episode 0's
        self.deterministic_replay = bool(deterministic_replay)
        cfg = IcemConfig(
            horizon=self.horizon, act_dim=self.dim_samples[1], num_traj=self.num_sim_traj,
            elites_size=self.elites_size, opt_iters=self.opt_iter, cost_mode=self.cost_along_trajectory,
            use_mean_actions=bool(self.use_mean_actions), keep_previous_elites=bool(self.keep_previous_elites),
            shift_elites=bool(self.shift_elites_over_time), factor_decrease=float(self.factor_decrease_num),
            alpha=float(self.alpha), init_std=float(self.init_std), fraction_reused=float(self.fraction_elites_reused),
            noise_beta=float(self.noise_beta), dtype=dtype, rng_rounds=rng_rounds, seed=seed, rank=rank, world=world)
        if cfg.cost_mode not in ("sum", "best", "final"):
            raise NotImplementedError(
                "Implement method {} to compute cost along trajectory".format(cfg.cost_mode))
        self.planner = IcemPlanner(cfg, self.env.action_space.low, self.env.action_space.high, device=device,
                                   process_group=process_group)
        self._bind_models(world)
        self._elite_costs = None
        self._elite_actions = None
        self.last_min_cost = None

    # -- parameter handling: icem.py:213-247 ---------------------------------------------------
    def _parse_action_sampler_params(self, *, alpha, elites_size, opt_iterations, init_std, use_mean_actions,
                                     keep_previous_elites, shift_elites_over_time, fraction_elites_reused,
                                     noise_beta=1):
        self.alpha = alpha
        self.elites_size = elites_size
        self.opt_iter = opt_iterations
        self.init_std = init_std
        self.use_mean_actions = use_mean_actions
        self.keep_previous_elites = keep_previous_elites
        self.shift_elites_over_time = shift_elites_over_time
        self.fraction_elites_reused = fraction_elites_reused
        self.noise_beta = noise_beta

    def _check_validity_parameters(self):
        self.num_elites = min(self.elites_size, self.num_sim_traj // 2)
        if self.num_elites < 2:
            warn('Number of trajectories is too low for given elites_frac. Setting num_elites to 2.')
            self.num_elites = 2
        space = self.env.action_space
        if isinstance(space, Discrete) or type(space).__name__ == "Discrete":
            raise NotImplementedError("CEM ERROR: Implement categorical distribution for discrete envs.")
        if isinstance(space, Box) or type(space).__name__ == "Box":
            self.dim_samples = (self.horizon, space.shape[0])
        else:
            raise NotImplementedError

    # -- state views ---------------------------------------------------------------------------
    @property
    def mean(self) -> np.ndarray:
        return self.planner.mean.detach().cpu().numpy().astype(np.float64)

    @property
    def std(self) -> np.ndarray:
        return self.planner.std.detach().cpu().numpy().astype(np.float64)

    @property
    def elite_samples(self) -> TrajectoryBatch:
        """Actions (and costs) of the current elite set, best first -- materialised on demand."""
        if self.device_path:
            if self.planner.mpc_step == 0:
                return TrajectoryBatch()
            a, c = self.planner.current_elites()
        else:
            if self._elite_actions is None:
                return TrajectoryBatch()
            a, c = self._elite_actions, self._elite_costs
        return TrajectoryBatch(actions=a.detach().cpu().numpy().astype(np.float64),
                               costs=c.detach().cpu().numpy().astype(np.float64))

    # -- rollout hooks: icem.py:31-46 ----------------------------------------------------------
    def beginning_of_rollout(self, *, observation, state=None, mode):
        super().beginning_of_rollout(observation=observation, state=state, mode=mode)
        if not self.deterministic_replay:
            self.planner.new_episode()
        if self.device_path:
            self.planner.reset()
        else:
            p = self.planner
            self._mean = torch.empty((p.h, p.d), dtype=p.dt, device=p.device)
            self._std = torch.empty_like(self._mean)
            p.reset_distribution(self._mean, self._std)
            p.mean, p.std = self._mean, self._std
            p.mpc_step = 0
        self._elite_actions = None
        self._elite_costs = None
        self.was_reset = True
        self.model_evals_per_timestep = sum(
            max(self.elites_size * 2, int(self.num_sim_traj / (self.factor_decrease_num ** i)))
            for i in range(self.opt_iter)) * self.horizon
        if self.verbose:
            print(f"iCEM using {self.model_evals_per_timestep} evaluations per step "
                  f"and {self.model_evals_per_timestep / self.horizon} trajectories per step")

    def end_of_rollout(self, total_time, total_return, mode):
        super().end_of_rollout(total_time, total_return, mode)

    # -- noise sources -------------------------------------------------------------------------
    def _noise_fn(self):
        if callable(self.noise_source):
            return self.noise_source
        if self.noise_source == "numpy_legacy":
            d, F = self.dim_samples[1], self.horizon // 2 + 1
            if self.noise_beta <= 0:
                return lambda num: (np.random.randn(num, self.horizon, d), None)  # icem.py:77

            def legacy(num):  # the two global-stream draws of colorednoise (call site icem.py:73)
                return np.random.normal(size=(num, d, F)), np.random.normal(size=(num, d, F))
            return legacy
        if self.noise_source == "philox":
            return None
        raise ValueError(f"unknown noise_source {self.noise_source!r}")

    # -- one MPC step: icem.py:106-189 ---------------------------------------------------------
    def get_action(self, obs, state, mode="train"):
        if not self.was_reset:
            raise AttributeError("beginning_of_rollout() needs to be called before")
        if self.verbose:   # icem.py:112-115
            print(f"-------------------- {self.mean[0][0:6]}")
            if mode != "expert":
                self.check_model_consistency()
        self.forward_model_state = self.forward_model.got_actual_observation_and_env_state(
            observation=obs, env_state=state, model_state=self.forward_model_state)
        noise = self._noise_fn()
        if self.device_path and self.verbose:
            # icem.py:151-158: best / mean / worst cost and the best first action of every iteration -- the per-iteration
            # form of the step (one host synchronisation per iteration: a debugging mode, as upstream)
            executed_dev = self.planner.plan_step(obs, noise=noise, on_iteration=self._print_iteration)
            host = torch.cat([executed_dev, self.planner.best_cost]).cpu().numpy().astype(np.float64)
            executed_action, self.last_min_cost = host[:-1], float(host[-1])
        elif self.device_path and noise is None and self.planner.cfg.world == 1:
            executed_action, self.last_min_cost = self.planner.get_action_host(obs)   # one call, one synchronisation
        elif self.device_path:
            executed_dev = self.planner.plan_step(obs, noise=noise)
            host = torch.cat([executed_dev, self.planner.best_cost]).cpu().numpy().astype(np.float64)  # one D2H sync
            executed_action, self.last_min_cost = host[:-1], float(host[-1])
        else:
            executed_action = self._get_action_stagewise(obs, noise)
        if self._new_mean_is_overridden(MpcICemHip):
            # icem.py:168-171: the device epilogue has shifted the mean and KEPT its last row (the default
            # compute_new_mean); a subclass decides that row from the best trajectory's last predicted observation
            best = self.elite_samples.as_array("actions")[0]
            new_last = np.asarray(self.compute_new_mean(obs=self._last_predicted_observation(obs, best)), dtype=np.float64)
            p = self.planner
            p.mean[-1].copy_(torch.as_tensor(new_last.reshape(p.d), dtype=p.dt, device=p.device))
        self.logger.log(self.last_min_cost, key="Expected_trajectory_cost")
        if self.do_visualize_plan:  # icem.py:180-183
            bt = self.best_trajectory(obs)
            self.visualize_plan(obs=bt["observations"], state=self.forward_model_state, acts=bt["actions"])
        if self.forward_model_state is not None:  # stateful models advance with the executed action
            _, self.forward_model_state, _ = self.forward_model.predict(
                observations=obs, states=self.forward_model_state, actions=executed_action)
        return executed_action

    @staticmethod
    def get_action_batch(controllers, observations, states=None, mode="train"):
        """``get_action`` of several controllers of ONE configuration at once -- the reference's parallel episodes, each
        controller its own ``get_action`` (icem/misc/rollout_utils.py:46-58, 129-152) -- through ``icem_plan_step_batch``:
        every stage of the planning step is one launch for all of them.  Each controller ends in exactly the state its own
        ``get_action(observations[i], states[i])`` leaves (same executed action, bit for bit; same hooks).  Device path with
        device noise only; anything else: call ``get_action`` per controller."""
        ctrls = list(controllers)
        n = len(ctrls)
        states = [None] * n if states is None else list(states)
        for c in ctrls:
            if not c.was_reset:
                raise AttributeError("beginning_of_rollout() needs to be called before")
            if not c.device_path or c._noise_fn() is not None or c.planner.cfg.world != 1 or c.verbose:
                raise NotImplementedError("get_action_batch: device path, Philox noise, one GPU, not verbose")
        for c, ob, stt in zip(ctrls, observations, states):
            c.forward_model_state = c.forward_model.got_actual_observation_and_env_state(
                observation=ob, env_state=stt, model_state=c.forward_model_state)
        IcemPlanner.plan_step_batch([c.planner for c in ctrls], observations)
        host = torch.stack([torch.cat([c.planner.executed, c.planner.best_cost]) for c in ctrls]).cpu().numpy().astype(np.float64)  # one D2H sync
        # what icem_get_action reports for a solo step: non-finite costs out of a finite observation (ICEM_E_RANGE)
        from ._lib import IcemError, ICEM_E_RANGE
        for c, ob in zip(ctrls, observations):
            seen = c.planner.nonfinite_costs()
            fresh, c._nonfinite_seen = seen - getattr(c, "_nonfinite_seen", 0), seen
            if fresh and np.all(np.isfinite(np.asarray(ob, dtype=np.float64))):
                raise IcemError(ICEM_E_RANGE, f"{fresh} trajectories of this MPC step came back with a non-finite cost from a finite observation")
        out = []
        for i, (c, ob) in enumerate(zip(ctrls, observations)):
            executed_action, c.last_min_cost = host[i, :-1].copy(), float(host[i, -1])
            if c._new_mean_is_overridden(MpcICemHip):
                best = c.elite_samples.as_array("actions")[0]
                new_last = np.asarray(c.compute_new_mean(obs=c._last_predicted_observation(ob, best)), dtype=np.float64)
                p = c.planner
                p.mean[-1].copy_(torch.as_tensor(new_last.reshape(p.d), dtype=p.dt, device=p.device))
            c.logger.log(c.last_min_cost, key="Expected_trajectory_cost")
            if c.do_visualize_plan:
                bt = c.best_trajectory(ob)
                c.visualize_plan(obs=bt["observations"], state=c.forward_model_state, acts=bt["actions"])
            if c.forward_model_state is not None:
                _, c.forward_model_state, _ = c.forward_model.predict(observations=ob, states=c.forward_model_state, actions=executed_action)
            out.append(executed_action)
        return out

    def compute_new_mean(self, obs):
        """icem.py:191-192: the last row of the shifted mean (``obs``: the best trajectory's last predicted observation).
        The default keeps the last row -- which is what the device epilogue (``icem_shift``) computes, so nothing is
        fetched; a subclass that overrides this is called once per ``get_action`` with that observation (re-rolled on
        demand) and its row is written into the device-resident mean."""
        return self.mean[-1]

    def _print_iteration(self, it):
        """The reference's verbose line (icem.py:151-158) from the device buffers of iteration ``it``: costs of the simulated
        pool (+ the kept elites behind it), the first action of the cheapest row."""
        p = self.planner
        n = p.population_sizes[it]
        n_sim = n + (p.n_reuse if (it == 0 and self.shift_elites_over_time and p.mpc_step > 0) else 0)
        costs = p.costs[:n_sim]
        if it > 0 and self.keep_previous_elites and p.n_reuse > 0:
            g = (p.mpc_step * p.cfg.opt_iters + it) & 1      # the set the merge of this iteration read
            costs = torch.cat([costs, p.elites_costs[g][:p.n_reuse]])
        c = costs.detach().cpu().numpy().astype(np.float64)
        best = int(np.argmin(c))
        first = (p.actions[best, 0] if best < n_sim else p.elites_actions[(p.mpc_step * p.cfg.opt_iters + it) & 1][best - n_sim, 0])
        scale = self.horizon if self.cost_along_trajectory == "sum" else 1.0
        print('iter {}:{} --- best cost: {:.2f} --- mean: {:.2f} --- worst: {:.2f}  best action: {}...'
              .format(it, n, np.amin(c) / scale, np.mean(c) / scale, np.amax(c) / scale,
                      first.detach().cpu().numpy().astype(np.float64)[0:6]))

    def best_trajectory(self, obs) -> TrajectoryBatch:
        """The best trajectory of the last CEM iteration as the reference hands it to ``visualize_plan`` and hooks
        (``simulated_paths[best_traj_idx]``, icem.py:180-183): a one-row batch with ``actions [1,h,d]``,
        ``observations`` / ``next_observations [1,h,o]`` and its cost.  The N rollouts of the planning step are never
        materialised (the reference builds N Rollout objects per iteration, ~90 % of its step time); this one is
        re-rolled from its action sequence on demand."""
        es = self.elite_samples
        if len(es) == 0:
            return TrajectoryBatch()
        acts = es.as_array("actions")[:1]
        if self.device_path:
            p = self.planner
            cost, o = p.rollout_cost(np.asarray(obs, dtype=np.float64), torch.as_tensor(acts, dtype=p.dt, device=p.device),
                                     return_observations=True)
            o = o.detach().cpu().numpy().astype(np.float64)          # [1, h, o]: observation BEFORE each action
            # the state BEHIND the last action comes from the device too: a second rollout that starts from the last
            # observation with that action first -- its observation before step 1 is the transition asked for
            if p.h >= 2:
                tail = np.zeros_like(acts)
                tail[:, 0] = acts[:, -1]
                _, o2 = p.rollout_cost(o[0, -1], torch.as_tensor(tail, dtype=p.dt, device=p.device), return_observations=True)
                nxt = o2.detach().cpu().numpy().astype(np.float64)[:, 1]
            else:   # (a one-step horizon has no second observation to read: the model's reference-style predict)
                nxt, _, _ = self.forward_model.predict(observations=o[:, -1], states=None, actions=acts[:, -1])
            next_o = np.concatenate([o[:, 1:], nxt[:, None]], axis=1)
            return TrajectoryBatch(observations=o, next_observations=next_o, actions=acts,
                                   costs=cost.detach().cpu().numpy().astype(np.float64))
        batch = self.simulate_trajectories(obs=obs, state=self.forward_model_state, action_sequences=acts)
        return batch

    def _get_action_stagewise(self, obs, noise):
        p = self.planner
        K, it_n = self.num_elites, self.opt_iter
        call_base = p.noise_offset(p.mpc_step * (it_n + 1))
        pool = costs_dev = idx = None
        for i, n_i in enumerate(p.population_sizes):
            z = noise(n_i) if noise is not None else (None, None)
            with_shift = i == 0 and self.shift_elites_over_time and self._elite_actions is not None and p.n_reuse > 0
            # (the shifted elites are sampled into the tail of the same buffer: nothing is concatenated)
            actions = torch.empty((n_i + (p.n_reuse if with_shift else 0), p.h, p.d), dtype=p.dt, device=p.device)
            p.sample_clip(n_i, p.mean, p.std, z[0], z[1], offset=call_base + i,
                          row0_mean=bool(self.use_mean_actions and i == it_n - 1), out=actions[:n_i])
            if with_shift:
                zs = noise(p.n_reuse) if noise is not None else (None, None)
                shifted = actions[n_i:]
                shifted[:, :-1] = self._elite_actions[:p.n_reuse, 1:]
                p.sample_clip(p.n_reuse, p.mean, p.std, zs[0], zs[1], offset=call_base + it_n, t_begin=p.h - 1,
                              out=shifted)
            costs = self._costs_of(obs, actions)
            pool = actions
            keep = i > 0 and self.keep_previous_elites and p.n_reuse > 0
            if p.can_update_in_one_launch(actions.shape[0] + (p.n_reuse if keep else 0), K):
                # top-K over [pool | kept elites] + gather + refit in one launch, nothing concatenated
                costs_dev, idx, self._elite_actions = p.update_distribution(
                    costs, actions, K, p.mean, p.std,
                    self._elite_costs[:p.n_reuse] if keep else None, self._elite_actions[:p.n_reuse] if keep else None)
            else:
                if keep:
                    pool = torch.cat([actions, self._elite_actions[:p.n_reuse]], dim=0)
                    costs = torch.cat([costs, self._elite_costs[:p.n_reuse]])
                costs_dev, idx = p.topk_sorted(costs, K)
                self._elite_actions = p.gather_refit(pool, idx, p.mean, p.std)
            self._elite_costs = costs_dev
        p.shift(p.mean, p.std)
        p.mpc_step += 1
        # best of the last pool (icem.py:163) and its cost: one device-to-host copy, one synchronisation
        host = torch.cat([self._elite_actions[0, 0], costs_dev[:1]]).cpu().numpy().astype(np.float64)
        self.last_min_cost = float(host[-1])
        return host[:-1]


# ---------------------------------------------------------------------------------------------
# the CEM baseline on the same kernels
# ---------------------------------------------------------------------------------------------

class MpcCemStdHip(MpcController):
    """``controllers.mpc.MpcCemStd`` (icem/controllers/mpc.py:142-327) -- the truncated-normal CEM the paper compares
    against -- with sampling, rollout (built-in model), top-K, refit and bounds on the device.  Same constructor as the
    reference; extra optional keywords ``dtype``, ``seed``, ``rng_rounds``, ``device`` and ``noise_source``
    ("philox": device uniforms; "numpy_legacy": scipy's draws from the global ``np.random`` stream -- parity mode; or
    a callable ``uniforms(num) -> [num, h, d]``)."""

    def __init__(self, *, action_sampler_params, dtype="f32", seed=0, rng_rounds=10, device="cuda:0",
                 noise_source: Union[str, Callable] = "philox", deterministic_replay=False, **kwargs):
        super().__init__(**kwargs)
        self._parse_action_sampler_params(**dict(action_sampler_params))
        self._check_validity_parameters()
        self.logger = _get_logger(self.__class__.__name__)
        self.was_reset = False
        self.noise_source = noise_source
        self.deterministic_replay = bool(deterministic_replay)
        if self.cost_along_trajectory not in ("sum", "best", "final"):
            raise NotImplementedError(
                "Implement method {} to compute cost along trajectory".format(self.cost_along_trajectory))
        cfg = IcemConfig(horizon=self.horizon, act_dim=self.dim_samples[1], num_traj=self.num_sim_traj,
                         elites_size=self.elites_size, opt_iters=self.opt_iter, cost_mode=self.cost_along_trajectory,
                         use_mean_actions=False, keep_previous_elites=False, shift_elites=False, factor_decrease=1.0,
                         alpha=float(self.alpha), init_std=float(self.init_std), dtype=dtype, rng_rounds=rng_rounds, seed=seed)
        self.planner = IcemPlanner(cfg, self.env.action_space.low, self.env.action_space.high, device=device)
        self._bind_models()
        self.last_min_cost = None

    # mpc.py:303-327
    def _parse_action_sampler_params(self, *, alpha, elites_size, opt_iterations, init_std, shift_means,
                                     execute_best_elite, bounds_like_levine):
        self.alpha = alpha
        self.elites_size = elites_size
        self.opt_iter = opt_iterations
        self.init_std = init_std
        self.execute_best_elite = execute_best_elite
        self.like_levine = bounds_like_levine
        self.shift_means = shift_means

    _check_validity_parameters = MpcICemHip._check_validity_parameters

    @property
    def mean(self) -> np.ndarray:
        return self._mean.detach().cpu().numpy().astype(np.float64)

    @property
    def std(self) -> np.ndarray:
        return self._std.detach().cpu().numpy().astype(np.float64)

    @property
    def elite_samples(self) -> TrajectoryBatch:
        if getattr(self, "_elite_actions", None) is None:
            return TrajectoryBatch()
        return TrajectoryBatch(actions=self._elite_actions.detach().cpu().numpy().astype(np.float64),
                               costs=self._elite_costs.detach().cpu().numpy().astype(np.float64))

    def compute_new_mean(self, obs):
        """mpc.py:265-269: zeros under ``like_levine``, else the (already shifted) mean's last row."""
        return self.mean[-1] * 0 if self.like_levine else self.mean[-1]

    def _reset_std(self):  # get_init_std(True), mpc.py:180-185, then _update_bounds
        p = self.planner
        scratch = torch.empty_like(self._mean)
        p.reset_distribution(scratch, self._std)
        self._lower, self._upper = p.cem_bounds(self._mean, self._std, self.like_levine)

    def beginning_of_rollout(self, *, observation, state=None, mode):  # mpc.py:158-170
        super().beginning_of_rollout(observation=observation, state=state, mode=mode)
        p = self.planner
        if not self.deterministic_replay:
            p.new_episode()
        self._mean = torch.empty((p.h, p.d), dtype=p.dt, device=p.device)
        self._std = torch.empty_like(self._mean)
        p.reset_distribution(self._mean, self._std)
        self._lower, self._upper = p.cem_bounds(self._mean, self._std, self.like_levine)
        self._elite_actions = self._elite_costs = None
        p.mpc_step = 0
        self.was_reset = True
        self.model_evals_per_timestep = self.num_sim_traj * self.opt_iter * self.horizon
        if self.verbose:
            print(f"CEM-Standard using {self.model_evals_per_timestep} evaluations per step "
                  f"and {self.model_evals_per_timestep / self.horizon} trajectories per step")

    def _uniform_fn(self):
        if callable(self.noise_source):
            return self.noise_source
        if self.noise_source == "numpy_legacy":  # what scipy.stats.truncnorm.rvs draws (mpc.py:196-197)
            return lambda num: np.random.uniform(size=(num, self.horizon, self.dim_samples[1]))
        if self.noise_source == "philox":
            return None
        raise ValueError(f"unknown noise_source {self.noise_source!r}")

    def get_action(self, obs, state, mode="train"):  # mpc.py:200-262
        if not self.was_reset:
            raise AttributeError("beginning_of_rollout() needs to be called before")
        self.forward_model_state = self.forward_model.got_actual_observation_and_env_state(
            observation=obs, env_state=state, model_state=self.forward_model_state)
        p, uniforms = self.planner, self._uniform_fn()
        actions = costs_sorted = idx = None
        for i in range(self.opt_iter):
            u = uniforms(self.num_sim_traj) if uniforms is not None else None
            actions = p.sample_truncnorm(self.num_sim_traj, self._mean, self._std, self._lower, self._upper, u,
                                         offset=p.noise_offset(p.mpc_step * self.opt_iter + i))
            costs = self._costs_of(obs, actions)
            if p.can_update_in_one_launch(actions.shape[0], self.num_elites):  # mpc.py:270-281 in one launch
                costs_sorted, idx, self._elite_actions = p.update_distribution(costs, actions, self.num_elites, self._mean, self._std)
            else:
                costs_sorted, idx = p.topk_sorted(costs, self.num_elites)      # mpc.py:270
                self._elite_actions = p.gather_refit(actions, idx, self._mean, self._std)  # mpc.py:271-281
            self._elite_costs = costs_sorted
            self._lower, self._upper = p.cem_bounds(self._mean, self._std, self.like_levine)
        if self.execute_best_elite:                                        # mpc.py:230-233
            executed_action = self._elite_actions[0, 0].cpu().numpy().astype(np.float64)
        else:
            executed_action = self._mean[0].cpu().numpy().astype(np.float64)
        if self.shift_means:                                               # mpc.py:236-243
            last = torch.zeros_like(self._mean[-1]) if self.like_levine else self._mean[-1].clone()
            self._mean[:-1] = self._mean[1:].clone()
            self._mean[-1] = last                                          # the default compute_new_mean (mpc.py:265-269)
            if self._new_mean_is_overridden(MpcCemStdHip):
                best = self._elite_actions[0].detach().cpu().numpy().astype(np.float64)
                new_last = np.asarray(self.compute_new_mean(obs=self._last_predicted_observation(obs, best)), dtype=np.float64)
                self._mean[-1].copy_(torch.as_tensor(new_last.reshape(p.d), dtype=p.dt, device=p.device))
        else:
            self._mean.zero_()
        self._reset_std()                                                  # mpc.py:244-245
        self.last_min_cost = float(costs_sorted[0])
        self.logger.log(self.last_min_cost / self.horizon if self.cost_along_trajectory == "sum" else self.last_min_cost,
                        key="Expected_trajectory_cost")
        p.mpc_step += 1
        if self.forward_model_state is not None:
            _, self.forward_model_state, _ = self.forward_model.predict(
                observations=obs, states=self.forward_model_state, actions=executed_action)
        return executed_action


# ---------------------------------------------------------------------------------------------
# the random-shooting baseline
# ---------------------------------------------------------------------------------------------

class MpcRandomHip(MpcController):
    """``controllers.mpc.MpcRandom`` (icem/controllers/mpc.py:86-138): uniform action sequences held for
    ``action_change_frequency`` steps, rolled out, the first action of the cheapest one executed -- sampling, rollout,
    cost and argmin on the device.  ``action_sampler_params`` needs ``action_change_frequency`` (attribute or key).
    Extra optional keywords: ``dtype``, ``seed``, ``rng_rounds``, ``device``, ``noise_source`` ("philox": device
    uniforms keyed by the block index; "action_space": ``env.action_space``-style draws ``np.random.random_sample(d)``
    in the reference's call order, two of them at construction -- parity mode; or a callable
    ``uniforms(first_block, n_blocks) -> [n_blocks, d]``)."""

    def __init__(self, *, action_sampler_params, dtype="f32", seed=0, rng_rounds=10, device="cuda:0",
                 noise_source: Union[str, Callable] = "philox", **kwargs):
        super().__init__(**kwargs)
        asp = action_sampler_params
        self.action_change_frequency = int(asp["action_change_frequency"] if isinstance(asp, Mapping)
                                           else asp.action_change_frequency)
        assert self.action_change_frequency < self.horizon  # mpc.py:92
        space = self.env.action_space
        if isinstance(space, Discrete) or type(space).__name__ == "Discrete":
            raise NotImplementedError("discrete action spaces are not supported by the device sampler")
        if self.cost_along_trajectory not in ("sum", "best", "final"):
            raise NotImplementedError(
                "Implement method {} to compute cost along trajectory".format(self.cost_along_trajectory))
        self.noise_source = noise_source
        d = int(np.prod(space.shape))
        cfg = IcemConfig(horizon=self.horizon, act_dim=d, num_traj=self.num_sim_traj, elites_size=2, opt_iters=1,
                         cost_mode=self.cost_along_trajectory, dtype=dtype, rng_rounds=rng_rounds, seed=seed)
        self.planner = IcemPlanner(cfg, space.low, space.high, device=device)
        self._bind_models()
        self.calls = 0            # MpcRandom.sample() calls so far (its counter runs on across MPC steps)
        self._draws = []          # parity mode: the unit draws of blocks 0, 1, ... as they are consumed
        if noise_source == "action_space":
            # RndController.__init__ draws previous_action (unused by MpcRandom), then MpcRandom its current_action
            np.random.random_sample(d)
            self._draws.append(np.random.random_sample(d))
        self.last_min_cost = None
        self.best_traj_idx = None

    def beginning_of_rollout(self, *, observation, state=None, mode):
        super().beginning_of_rollout(observation=observation, state=state, mode=mode)

    def _block_uniforms(self, first_block: int, n_blocks: int):
        if callable(self.noise_source):
            return np.asarray(self.noise_source(first_block, n_blocks))
        if self.noise_source == "action_space":
            d = self.planner.d
            while len(self._draws) < first_block + n_blocks:
                self._draws.append(np.random.random_sample(d))
            return np.stack(self._draws[first_block:first_block + n_blocks])
        if self.noise_source == "philox":
            return None
        raise ValueError(f"unknown noise_source {self.noise_source!r}")

    def sample_action_sequences(self, obs, num_traj, time_slice=None) -> torch.Tensor:  # mpc.py:104-109
        p, f = self.planner, self.action_change_frequency
        block = lambda c: 0 if c < f else 1 + (c - f) // (f + 1)  # noqa: E731
        first, last = block(self.calls), block(self.calls + num_traj * p.h - 1)
        u = self._block_uniforms(first, last - first + 1)
        actions = p.sample_piecewise(num_traj, self.calls, f, u, first)
        self.calls += num_traj * p.h
        return actions

    def get_action(self, obs, state, mode="train"):  # mpc.py:114-138
        self.forward_model_state = self.forward_model.got_actual_observation_and_env_state(
            observation=obs, env_state=state, model_state=self.forward_model_state)
        p = self.planner
        actions = self.sample_action_sequences(obs, self.num_sim_traj)
        costs = self._costs_of(obs, actions)
        best_cost, idx = p.topk_sorted(costs, 1)                         # np.argmin(costs), mpc.py:122
        self.best_traj_idx = int(idx[0])
        self.last_min_cost = float(best_cost[0])
        self._last_actions, self._last_costs = actions, costs
        executed_action = actions[self.best_traj_idx, 0].cpu().numpy().astype(np.float64)
        if self.forward_model_state is not None:
            _, self.forward_model_state, _ = self.forward_model.predict(
                observations=obs, states=self.forward_model_state, actions=executed_action)
        return executed_action


# ---------------------------------------------------------------------------------------------
# registry: icem/controllers/__init__.py:6-31
# ---------------------------------------------------------------------------------------------

def controller_from_string(controller_str):
    return ControllerFactory(controller_str=controller_str)


class ControllerFactory:
    valid_base_controllers = {
        "mpc-icem-hip": (".controllers", "MpcICemHip"),
        "mpc-icem": (".controllers", "MpcICemHip"),
        "mpc-cem-std-hip": (".controllers", "MpcCemStdHip"),
        "mpc-cem-std": (".controllers", "MpcCemStdHip"),
        "mpc-random-hip": (".controllers", "MpcRandomHip"),
        "mpc-random": (".controllers", "MpcRandomHip"),
    }
    controller = None

    def __new__(cls, *, controller_str):
        if controller_str in cls.valid_base_controllers:
            pkg, name = cls.valid_base_controllers[controller_str]
            cls.controller = getattr(import_module(pkg, "icem_amd"), name)
        else:
            raise ImportError(f"cannot find '{controller_str}' in known controller: "
                              f"{cls.valid_base_controllers.keys()}")
        return cls.controller
