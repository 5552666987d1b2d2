"""Synthetic environment objects exposing what the hot path reads from an env:
``name``, ``action_space.low/high/shape`` (icem/controllers/icem.py:50,57,79,245) and
``cost_fn(obs, act, next_obs)`` (icem/controllers/abstract_controller.py:70,79).

MuJoCo is CPU-only and absent from the image, so the two shipped cost functions are restated
here with the HalfCheetah / HumanoidStandup shapes; ``cost_spec`` is the parametric form the
HIP rollout kernel evaluates on the device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


class Box:
    """Minimal stand-in for ``gym.spaces.Box`` (float32 bounds like gym)."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(np.float32)


class Discrete:
    def __init__(self, n):
        self.n = n


@dataclass
class CostSpec:
    """The parametric cost the device evaluates (``icem_cost_spec`` + ``icem_cost_terms``, include/icem_hip.h):
    ``flip penalties + ctrl_weight*sum(a^2) + lin_weight*obs[lin_idx] + diff_weight*(next_obs[diff_idx] - obs[diff_idx])
    + health_penalty*unhealthy(obs) + sum_j dist_weight[j]*f_j(||obs[a_j:a_j+len_j] - obs[b_j:b_j+len_j]||)``."""
    ctrl_weight: float = 0.1
    lin_idx: int = 8
    lin_weight: float = -1.0
    flip_idx: int = 1
    flip_penalty: float = 10.0
    flip_thresh: float = math.pi / 2
    diff_idx: int = -1
    diff_weight: float = 0.0
    health_idx: int = -1
    health_penalty: float = 0.0
    health_lo: float = 0.0
    health_hi: float = 0.0
    health_closed: bool = False
    box_from: int = -1
    box_lo: float = -100.0
    box_hi: float = 100.0
    dist_a: tuple = (0, 0)
    dist_b: tuple = (-1, -1)
    dist_len: tuple = (0, 0)
    dist_sparse: tuple = (False, False)
    dist_weight: tuple = (0.0, 0.0)
    dist_thresh: tuple = (0.0, 0.0)

    @property
    def extended(self) -> bool:
        return self.diff_idx >= 0 or self.health_idx >= 0 or any(n > 0 for n in self.dist_len)


class SyntheticEnv:
    """An env object carrying what the hot path reads: ``name``, ``action_space`` and ``cost_fn`` -- the reference's
    cost functions restated on ``cost_spec`` (HalfCheetah / HumanoidStandup score the PRE-action observation and
    ignore ``next_obs`` exactly as in the reference; Ant / Hopper read ``next_obs[0]``)."""

    def __init__(self, name: str, obs_dim: int, low, high, cost_spec: CostSpec):
        self.name = name
        self.obs_dim = obs_dim
        self.action_space = Box(low, high)
        self.cost_spec = cost_spec

    def unhealthy_states(self, observation):
        c = self.cost_spec
        z = observation[..., c.health_idx]
        ok = (c.health_lo <= z) * (z <= c.health_hi) if c.health_closed else (c.health_lo < z) * (z < c.health_hi)
        if c.box_from >= 0:
            st = observation[..., c.box_from:]
            ok = np.logical_and(np.all(np.logical_and(c.box_lo < st, st < c.box_hi), axis=-1), ok)
        return 1 - np.isfinite(observation).all(axis=-1) * ok

    def cost_fn(self, observation, action, next_obs=None):
        c = self.cost_spec
        observation = np.asarray(observation)
        action = np.asarray(action)
        scores = np.zeros(action.shape[:-1])
        if c.flip_idx >= 0:
            ang = observation[..., c.flip_idx]
            scores = scores + (ang > c.flip_thresh) * c.flip_penalty
            scores = scores + (ang < -c.flip_thresh) * c.flip_penalty
        scores = scores + c.ctrl_weight * np.sum(action ** 2, axis=-1)
        scores = scores + c.lin_weight * observation[..., c.lin_idx]
        if c.diff_idx >= 0:
            scores = scores + c.diff_weight * (np.asarray(next_obs)[..., c.diff_idx] - observation[..., c.diff_idx])
        if c.health_idx >= 0:
            scores = scores + c.health_penalty * self.unhealthy_states(observation)
        for j in range(2):
            if c.dist_len[j] > 0:
                v = observation[..., c.dist_a[j]:c.dist_a[j] + c.dist_len[j]]
                if c.dist_b[j] >= 0:
                    v = v - observation[..., c.dist_b[j]:c.dist_b[j] + c.dist_len[j]]
                r = np.linalg.norm(v, axis=-1)
                if c.dist_sparse[j]:
                    r = np.asarray(r > c.dist_thresh[j], dtype=np.float64)
                scores = scores + c.dist_weight[j] * r
        return scores

    def reward_fn(self, observation, action, next_obs=None):
        return -self.cost_fn(observation, action, next_obs)


def halfcheetah_env(obs_dim: int = 17, penalise_flipping: bool = True) -> SyntheticEnv:
    """HalfCheetah shapes: d=6, bounds +-1; angle/velocity at [1]/[8] (o=17) or [2]/[9] (o=18)
    -- icem/environments/mujoco.py:77-82; flipping penalty per settings/halfcheetah_running."""
    if obs_dim == 18:
        ang, vel = 2, 9
    elif obs_dim == 17:
        ang, vel = 1, 8
    else:
        raise ValueError(f'Got state of dimension {obs_dim}. Possible dimensions are 17 or 18.')
    spec = CostSpec(0.1, vel, -1.0, ang if penalise_flipping else -1, 10.0, math.pi / 2)
    return SyntheticEnv("HalfCheetah", obs_dim, -np.ones(6), np.ones(6), spec)


def humanoid_standup_env(obs_dim: int = 24) -> SyntheticEnv:
    """HumanoidStandup action shapes: d=17, bounds +-0.4; cost -obs[2] + 0.1*sum(a^2)
    (icem/environments/mujoco.py:259-277).  ``obs_dim`` is a synthetic latent size (the real
    env has o=378; only obs[2] enters the cost)."""
    spec = CostSpec(0.1, 2, -1.0, -1, 0.0, math.pi / 2)
    return SyntheticEnv("HumanoidStandup", obs_dim, -0.4 * np.ones(17), 0.4 * np.ones(17), spec)


def ant_env(dt: float = 0.05, ctrl_cost_weight: float = 0.5, healthy_z_range=(0.2, 1.0)) -> SyntheticEnv:
    """Ant shapes (o=113 with positions, d=8, bounds +-1): ``-(x' - x)/dt + 100*unhealthy + w*sum(a^2)``
    (icem/environments/mujoco.py:146-171)."""
    spec = CostSpec(ctrl_cost_weight, 0, 0.0, -1, 0.0, 0.0, diff_idx=0, diff_weight=-1.0 / dt, health_idx=2,
                    health_penalty=100.0, health_lo=healthy_z_range[0], health_hi=healthy_z_range[1], health_closed=True)
    return SyntheticEnv("Ant", 113, -np.ones(8), np.ones(8), spec)


def hopper_env(dt: float = 0.008, ctrl_cost_weight: float = 1e-3, healthy_z_range=(0.7, float("inf")),
               healthy_state_range=(-100.0, 100.0)) -> SyntheticEnv:
    """Hopper shapes (o=12 with positions, d=3): ``-(x' - x)/dt + 200*unhealthy + w*sum(a^2)``
    (icem/environments/mujoco.py:189-225; healthy_angle is dropped by the reference's own ``logical_and(.., out)``)."""
    spec = CostSpec(ctrl_cost_weight, 0, 0.0, -1, 0.0, 0.0, diff_idx=0, diff_weight=-1.0 / dt, health_idx=1,
                    health_penalty=200.0, health_lo=healthy_z_range[0], health_hi=healthy_z_range[1],
                    box_from=2, box_lo=healthy_state_range[0], box_hi=healthy_state_range[1])
    return SyntheticEnv("Hopper", 12, -np.ones(3), np.ones(3), spec)


def humanoid_env(obs_dim: int = 376, nq: int = 24, exclude_current_positions: bool = True,
                 forward_reward_weight: float = 1.25, ctrl_cost_weight: float = 0.1,
                 healthy_z_range=(1.0, 2.0)) -> SyntheticEnv:
    """Humanoid shapes (d=17, bounds +-0.4): ``-w_f*obs[nq-2 | nq] + 100*unhealthy(z = obs[0 | 2]) + w*sum(a^2)``
    (icem/environments/mujoco.py:302-343)."""
    spec = CostSpec(ctrl_cost_weight, nq - 2 if exclude_current_positions else nq, -forward_reward_weight, -1, 0.0, 0.0,
                    health_idx=0 if exclude_current_positions else 2, health_penalty=100.0,
                    health_lo=healthy_z_range[0], health_hi=healthy_z_range[1])
    return SyntheticEnv("Humanoid", obs_dim, -0.4 * np.ones(17), 0.4 * np.ones(17), spec)


def reacher_env(obs_dim: int = 11) -> SyntheticEnv:
    """Reacher shapes (d=2): ``||obs[-3:]||`` (icem/environments/mujoco.py:366-368)."""
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, dist_a=(obs_dim - 3, 0), dist_len=(3, 0), dist_weight=(1.0, 0.0))
    return SyntheticEnv("Reacher", obs_dim, -np.ones(2), np.ones(2), spec)


def fetch_pick_and_place_env(orig_obs_len: int = 25, sparse: bool = False, threshold: float = 0.05,
                             shaped_reward: bool = True) -> SyntheticEnv:
    """FetchPickAndPlace shapes (o = 25 + 3 goal entries, d=4): ``||goal - obs[3:6]|| + 0.1*||obs[0:3] - obs[3:6]||``
    or the ``[. > threshold]`` indicators (icem/environments/robotics.py:150-164)."""
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, dist_a=(orig_obs_len, 0), dist_b=(3, 3),
                    dist_len=(3, 3 if shaped_reward else 0), dist_sparse=(sparse, sparse), dist_weight=(1.0, 0.1),
                    dist_thresh=(threshold, threshold))
    return SyntheticEnv("FetchPickAndPlace", orig_obs_len + 3, -np.ones(4), np.ones(4), spec)


def fetch_reach_env(orig_obs_len: int = 10, sparse: bool = False, threshold: float = 0.05) -> SyntheticEnv:
    """FetchReach shapes (o = 10 + 3, d=4): ``||goal - obs[0:3]||`` or ``[. > threshold]`` (robotics.py:286-295)."""
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, dist_a=(orig_obs_len, 0), dist_b=(0, -1), dist_len=(3, 0),
                    dist_sparse=(sparse, False), dist_weight=(1.0, 0.0), dist_thresh=(threshold, 0.0))
    return SyntheticEnv("FetchReach", orig_obs_len + 3, -np.ones(4), np.ones(4), spec)
