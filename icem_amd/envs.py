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
    ctrl_weight: float = 0.1
    lin_idx: int = 8
    lin_weight: float = -1.0
    flip_idx: int = 1
    flip_penalty: float = 10.0
    flip_thresh: float = math.pi / 2


class SyntheticEnv:
    """``cost_t = ctrl_weight*sum(a^2) + lin_weight*obs[lin_idx] + flip penalties`` on the
    PRE-action observation; ``next_obs`` is ignored exactly as in the reference."""

    def __init__(self, name: str, obs_dim: int, low, high, cost_spec: CostSpec):
        self.name = name
        self.obs_dim = obs_dim
        self.action_space = Box(low, high)
        self.cost_spec = cost_spec

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
