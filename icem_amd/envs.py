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


TERM_NORM, TERM_NORM_GT, TERM_NORM_LT, TERM_SQ_OFFSET, TERM_SUMSQ, TERM_STEP_GT = range(6)


@dataclass(frozen=True)
class CostTerm:
    """One ``icem_cost_term``: ``weight * [obs[gate_idx] > gate_thresh] * f`` with ``r = ||obs[a:a+len] - obs[b:b+len]||``
    (``b < 0``: the slice itself) and ``f`` = ``r`` (NORM), ``[r > thresh]`` (NORM_GT), ``[r < thresh]`` (NORM_LT),
    ``(obs[a] - thresh)^2`` (SQ_OFFSET), ``sum obs[a:a+len]^2`` (SUMSQ), ``[obs[a] > thresh]`` (STEP_GT)."""
    kind: int
    a: int
    b: int = -1
    len: int = 1
    weight: float = 1.0
    thresh: float = 0.0
    gate_idx: int = -1
    gate_thresh: float = 0.0


@dataclass
class CostSpec:
    """The parametric cost the device evaluates (``icem_cost_spec`` + ``icem_cost_terms``, include/icem_hip.h):
    ``flip penalties + ctrl_weight*sum(a^2) + lin_weight*obs[lin_idx] + diff_weight*(next_obs[diff_idx] - obs[diff_idx])
    + health_penalty*unhealthy(obs) + sum of the terms``."""
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
    terms: tuple = ()

    @property
    def extended(self) -> bool:
        return self.diff_idx >= 0 or self.health_idx >= 0 or len(self.terms) > 0


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
        if c.lin_weight != 0:
            scores = scores + c.lin_weight * observation[..., c.lin_idx]
        if c.diff_idx >= 0:
            scores = scores + c.diff_weight * (np.asarray(next_obs)[..., c.diff_idx] - observation[..., c.diff_idx])
        if c.health_idx >= 0:
            scores = scores + c.health_penalty * self.unhealthy_states(observation)
        for tm in c.terms:
            if tm.kind == TERM_STEP_GT:
                f = np.asarray(observation[..., tm.a] > tm.thresh, dtype=np.float64)
            elif tm.kind == TERM_SQ_OFFSET:
                f = (observation[..., tm.a] - tm.thresh) ** 2
            else:
                v = observation[..., tm.a:tm.a + tm.len]
                if tm.b >= 0:
                    v = v - observation[..., tm.b:tm.b + tm.len]
                if tm.kind == TERM_SUMSQ:
                    f = np.sum(v ** 2, axis=-1)
                else:
                    r = np.linalg.norm(v, axis=-1)
                    f = r if tm.kind == TERM_NORM else np.asarray(r > tm.thresh if tm.kind == TERM_NORM_GT else r < tm.thresh,
                                                                  dtype=np.float64)
            if tm.gate_idx >= 0:
                f = f * (observation[..., tm.gate_idx] > tm.gate_thresh)
            scores = scores + tm.weight * f
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
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=(CostTerm(TERM_NORM, obs_dim - 3, -1, 3),))
    return SyntheticEnv("Reacher", obs_dim, -np.ones(2), np.ones(2), spec)


def fetch_pick_and_place_env(orig_obs_len: int = 25, sparse: bool = False, threshold: float = 0.05,
                             shaped_reward: bool = True) -> SyntheticEnv:
    """FetchPickAndPlace shapes (o = 25 + 3 goal entries, d=4): ``||goal - obs[3:6]|| + 0.1*||obs[0:3] - obs[3:6]||``
    or the ``[. > threshold]`` indicators (icem/environments/robotics.py:150-164)."""
    kind = TERM_NORM_GT if sparse else TERM_NORM
    terms = (CostTerm(kind, orig_obs_len, 3, 3, 1.0, threshold),)
    if shaped_reward:
        terms += (CostTerm(kind, 0, 3, 3, 0.1, threshold),)
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms)
    return SyntheticEnv("FetchPickAndPlace", orig_obs_len + 3, -np.ones(4), np.ones(4), spec)


def fetch_reach_env(orig_obs_len: int = 10, sparse: bool = False, threshold: float = 0.05) -> SyntheticEnv:
    """FetchReach shapes (o = 10 + 3, d=4): ``||goal - obs[0:3]||`` or ``[. > threshold]`` (robotics.py:286-295)."""
    spec = CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0,
                    terms=(CostTerm(TERM_NORM_GT if sparse else TERM_NORM, orig_obs_len, 0, 3, 1.0, threshold),))
    return SyntheticEnv("FetchReach", orig_obs_len + 3, -np.ones(4), np.ones(4), spec)


def door_env(obs_dim: int = 39, nq: int = 30, nv: int = 30, shaped_reward: bool = True, add_bonus_rewards: bool = True) -> SyntheticEnv:
    """Door (hand manipulation suite) shapes (o=39, d=28): ``0.1*||palm - handle|| + 0.1*(door - 1.57)^2
    + 1e-5*sum(obs[-nv:]^2) - 2[door > 0.2] - 8[door > 1.0] - 10[door > 1.35]`` with door = obs[nq-2], palm =
    obs[nq-1:nq+2], handle = obs[nq+2:nq+5] (icem/environments/mjenvs.py:26-31, 57-78)."""
    door, palm, handle = nq - 2, nq - 1, nq + 2
    tail = min(nv, obs_dim)
    terms = ((CostTerm(TERM_NORM, palm, handle, 3, 0.1),) if shaped_reward else ()) + (
        CostTerm(TERM_SQ_OFFSET, door, -1, 1, 0.1, 1.57), CostTerm(TERM_SUMSQ, obs_dim - tail, -1, tail, 1e-5))
    if add_bonus_rewards:
        terms += (CostTerm(TERM_STEP_GT, door, -1, 1, -2.0, 0.2), CostTerm(TERM_STEP_GT, door, -1, 1, -8.0, 1.0),
                  CostTerm(TERM_STEP_GT, door, -1, 1, -10.0, 1.35))
    return SyntheticEnv("Door", obs_dim, -np.ones(28), np.ones(28), CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms))


def relocate_env(obs_dim: int = 39, nq: int = 36, add_bonus_rewards: bool = True) -> SyntheticEnv:
    """Relocate shapes (o=39, d=30): ``0.1*||palm - obj|| - [obj_z > 0.04] + 0.5*||obj - target||*[obj_z > 0.04]
    - 10[||obj - target|| < 0.1] - 20[||obj - target|| < 0.05]`` with the three difference vectors at obs[nq-6:nq+3]
    and obj_z = obs[-1] (icem/environments/mjenvs.py:112-115, 155-174)."""
    po, ot, z = nq - 6, nq, obs_dim - 1
    terms = (CostTerm(TERM_NORM, po, -1, 3, 0.1), CostTerm(TERM_STEP_GT, z, -1, 1, -1.0, 0.04),
             CostTerm(TERM_NORM, ot, -1, 3, 0.5, 0.0, z, 0.04))
    if add_bonus_rewards:
        terms += (CostTerm(TERM_NORM_LT, ot, -1, 3, -10.0, 0.1), CostTerm(TERM_NORM_LT, ot, -1, 3, -20.0, 0.05))
    return SyntheticEnv("Relocate", obs_dim, -np.ones(30), np.ones(30), CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms))
