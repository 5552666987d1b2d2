"""CPU oracle for the iCEM inner planning loop.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``icem_amd`` (the product) never imports anything under ``oracle/``.

It restates, in plain NumPy (float64 by default, like the reference), the
algorithm of the reference's hot path.  Every function cites the reference
file:line it follows (paths relative to ``/root/reference``):

* ``colorednoise.powerlaw_psd_gaussian`` -- third-party PyPI package
  ``colorednoise`` (F. Patzelt), **unpinned** in the reference
  (``Pipfile:10``: ``colorednoise = "*"``), source absent from
  ``/root/reference``.  Restated from its published 1.x algorithm; call site
  ``icem/controllers/icem.py:73-75``.
* ``MpcICem.sample_action_sequences``      ``icem/controllers/icem.py:61-82``
* ``MpcICem.prepare_action_sequences``     ``icem/controllers/icem.py:84-89``
* ``MpcICem.elites_2_action_sequences``    ``icem/controllers/icem.py:91-104``
* ``MpcICem.get_action``                   ``icem/controllers/icem.py:106-189``
* ``MpcICem.update_distributions``         ``icem/controllers/icem.py:194-211``
* ``ModelBasedController.trajectory_cost_fn``
                                           ``icem/controllers/abstract_controller.py:74-91``
* ``ForwardModelWithDefaults.predict_n_steps`` (batched rollout loop)
                                           ``icem/models/abstract_models.py:17-53``
* HalfCheetah / HumanoidStandup ``cost_fn`` ``icem/environments/mujoco.py:67-99, 259-277``

Pinning: the reference has no tests and no golden vectors for this path, and
``colorednoise`` is unpinned; the oracle is pinned against outputs of the
reference *itself*, imported in the build container (with stub modules for
its missing third-party imports) by ``tests/golden/make_golden.py``; the
resulting fixtures are ``tests/golden/*.npz`` and are checked by
``tests/test_oracle_golden.py``.  The MuJoCo ground-truth dynamics cannot run
here (mujoco-py absent): rollouts are pinned through the model *interface*
with synthetic batched models only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# a-1  colorednoise.powerlaw_psd_gaussian  (third-party; icem.py:73-75 call site)
# --------------------------------------------------------------------------


def psd_scale(h: int, beta: float) -> Tuple[np.ndarray, float]:
    """Per-frequency amplitude ``s_scale[F]`` and the normalising ``sigma``.

    Follows colorednoise 1.x ``powerlaw_psd_gaussian`` with ``fmin=0``:
    ``f = rfftfreq(h)``; frequencies below ``1/h`` (only DC) take the value of
    the first non-cut bin; ``s = f**(-beta/2)``; ``sigma`` is the theoretical
    output std.
    """
    f = np.fft.rfftfreq(h)
    s = f.copy()
    fmin = max(0.0, 1.0 / h)
    ix = int(np.sum(s < fmin))
    if ix and ix < len(s):
        s[:ix] = s[ix]
    s = s ** (-beta / 2.0)
    w = s[1:].copy()
    w[-1] *= (1 + (h % 2)) / 2.0
    sigma = 2.0 * math.sqrt(float(np.sum(w ** 2))) / h
    return s, sigma


def colored_from_white(beta: float, h: int, z_r: np.ndarray, z_i: np.ndarray) -> np.ndarray:
    """``powerlaw_psd_gaussian(beta, size=(..., h))`` given the white draws.

    ``z_r, z_i`` are the standard-normal draws ``[..., F]`` (F = h//2+1) that
    upstream obtains from two consecutive ``numpy.random.normal(scale=s_scale,
    size=[..., F])`` calls (``sr = z_r*s_scale`` bit-exactly).  Returns
    ``[..., h]`` with the temporal correlation along the last axis.
    """
    s, sigma = psd_scale(h, beta)
    sr = z_r * s
    si = z_i * s
    if not (h % 2):
        si[..., -1] = 0
    si[..., 0] = 0
    spec = sr + 1j * si
    return np.fft.irfft(spec, n=h, axis=-1) / sigma


def synthesis_matrices(h: int, beta: float) -> Tuple[np.ndarray, np.ndarray]:
    """Real synthesis matrices ``Cr, Ci`` of shape ``[F, h]`` (float64) with

        colored_from_white(beta, h, z_r, z_i) == z_r @ Cr + z_i @ Ci

    up to rounding: the inverse real DFT, ``s_scale`` and ``1/sigma`` folded
    into one table.  Rows ``Ci[0]`` (DC) and, for even ``h``, ``Ci[F-1]``
    (Nyquist) are zero -- those draws are discarded by upstream.
    """
    s, sigma = psd_scale(h, beta)
    F = h // 2 + 1
    k = np.arange(F, dtype=np.float64)[:, None]
    t = np.arange(h, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * k * t / h
    mult = np.full((F, 1), 2.0)
    mult[0, 0] = 1.0
    if h % 2 == 0:
        mult[F - 1, 0] = 1.0
    amp = mult * s[:, None] / (h * sigma)
    Cr = amp * np.cos(ang)
    Ci = -amp * np.sin(ang)
    Ci[0, :] = 0.0
    if h % 2 == 0:
        Ci[F - 1, :] = 0.0
    return Cr, Ci


def legacy_white_noise(num: int, d: int, h: int) -> Tuple[np.ndarray, np.ndarray]:
    """The two draws upstream takes from the *global legacy* ``np.random``
    stream for ``size=(num, d, h)`` (reference seeding: ``misc/seeding.py:13-19``)."""
    F = h // 2 + 1
    z_r = np.random.normal(size=(num, d, F))
    z_i = np.random.normal(size=(num, d, F))
    return z_r, z_i


# --------------------------------------------------------------------------
# Philox4x32-seeded xoshiro128++ + Box-Muller (the build's device RNG, restated)
# --------------------------------------------------------------------------

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SH = np.uint64(32)


def philox4x32(c0, c1, c2, c3, k0: int, k1: int, rounds: int = 10):
    """Philox4x32-R (Salmon et al. 2011).  Counter words are array-likes of
    uint32 values (held in uint64 arrays); key words are Python ints."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _SH, p0 & _MASK
        hi1, lo1 = p1 >> _SH, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def box_muller(xa: np.ndarray, xb: np.ndarray, dtype=np.float64):
    """Two normals from two uint32 words.  ``u1 = (xa+0.5)*2^-32`` (f64) or
    ``fma(xa, 2^-32, 2^-33)`` (f32); angle ``2*pi*xb*2^-32``; returns
    ``(r*cos, r*sin)``."""
    if dtype == np.float64:
        u1 = (xa.astype(np.float64) + 0.5) * (2.0 ** -32)
        v = xb.astype(np.float64) * (2.0 ** -32)
        r = np.sqrt(-2.0 * np.log(u1))
        ang = 2.0 * np.pi * v
        return r * np.cos(ang), r * np.sin(ang)
    xa32 = xa.astype(np.float32)
    xb32 = xb.astype(np.float32)
    u1 = xa32 * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)
    v = xb32 * np.float32(2.0 ** -32)
    r = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = np.float32(2.0 * np.pi) * v
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def xoshiro128pp_words(s0, s1, s2, s3, count: int):
    """``count`` successive outputs of xoshiro128++ (Blackman & Vigna) for arrays of uint32 states
    (held in uint64 arrays)."""
    M = _MASK
    out = []
    for _ in range(count):
        sm = (s0 + s3) & M
        res = ((((sm << np.uint64(7)) | (sm >> np.uint64(25))) & M) + s0) & M
        t = (s1 << np.uint64(9)) & M
        s2 = s2 ^ s0
        s3 = s3 ^ s1
        s1 = s1 ^ s2
        s0 = s0 ^ s3
        s2 = s2 ^ t
        s3 = ((s3 << np.uint64(11)) | (s3 >> np.uint64(21))) & M
        out.append(res)
    return out


def philox_white_noise(seed: int, offset: int, num: int, d: int, h: int,
                       first_index: int = 0, rounds: int = 10,
                       dtype=np.float64) -> Tuple[np.ndarray, np.ndarray]:
    """White draws ``z_r, z_i [num, d, F]`` from the build's device RNG.

    Row ``(n, j)`` (n = *global* trajectory index ``first_index + local``): ONE
    Philox4x32-``rounds`` call with counter ``(n, j<<16, offset_lo, offset_hi)`` and key
    ``(seed_lo, seed_hi)`` seeds a xoshiro128++ state; its successive words ``x_0, x_1, ...``
    give the normals ``m = 2i, 2i+1`` by Box-Muller on ``(x_{2i}, x_{2i+1})``.  Normal
    ``m < F`` is ``z_r[k=m]``; ``m >= F`` is ``z_i[k=m-F+1]``.  Unused ``z_i`` slots (DC,
    even-h Nyquist) are returned as 0.
    """
    F = h // 2 + 1
    n_idx = np.broadcast_to((first_index + np.arange(num, dtype=np.uint64))[:, None], (num, d))
    j_idx = np.broadcast_to((np.arange(d, dtype=np.uint64) << np.uint64(16))[None, :], (num, d))
    s0, s1, s2, s3 = philox4x32(n_idx, j_idx, offset & 0xFFFFFFFF, (offset >> 32) & 0xFFFFFFFF,
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds)
    npairs = (h + 1) // 2
    words = xoshiro128pp_words(s0, s1, s2, s3, 2 * npairs)
    g = np.empty((num, d, 2 * npairs), dtype=dtype)
    for i in range(npairs):
        g[..., 2 * i], g[..., 2 * i + 1] = box_muller(words[2 * i], words[2 * i + 1], dtype)
    g = g[..., :h]
    z_r = np.ascontiguousarray(g[..., :F])
    z_i = np.zeros((num, d, F), dtype=dtype)
    n_im = h - F
    z_i[..., 1:1 + n_im] = g[..., F:]
    return z_r, z_i


def philox_white_randn(seed: int, offset: int, num: int, d: int, h: int, first_index: int = 0, rounds: int = 10,
                       dtype=np.float64) -> np.ndarray:
    """The ``beta <= 0`` draw ``randn(num, h, d)`` (icem.py:77) from the device RNG: entry ``[n, t, j]`` is normal
    ``t`` of row ``(n, j)``'s stream (the same streams as :func:`philox_white_noise`)."""
    F = h // 2 + 1
    z_r, z_i = philox_white_noise(seed, offset, num, d, h, first_index, rounds, dtype)
    g = np.concatenate([z_r, z_i[..., 1:1 + (h - F)]], axis=-1)  # [num, d, h] in draw order
    return np.ascontiguousarray(g.transpose([0, 2, 1]))


def philox_uniforms(seed: int, offset: int, num: int, d: int, h: int, first_index: int = 0, rounds: int = 10,
                    dtype=np.float64) -> np.ndarray:
    """Uniform draws ``[num, h, d]`` of the device's truncated-normal sampler: entry ``[n, t, j]`` is word ``t`` of row
    ``(n, j)``'s stream (same streams as :func:`philox_white_noise`) mapped to ``(x + 0.5) * 2**-32``."""
    n_idx = np.broadcast_to((first_index + np.arange(num, dtype=np.uint64))[:, None], (num, d))
    j_idx = np.broadcast_to((np.arange(d, dtype=np.uint64) << np.uint64(16))[None, :], (num, d))
    s0, s1, s2, s3 = philox4x32(n_idx, j_idx, offset & 0xFFFFFFFF, (offset >> 32) & 0xFFFFFFFF,
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds)
    words = xoshiro128pp_words(s0, s1, s2, s3, h)
    u = np.stack([(w.astype(dtype) + dtype(0.5)) * dtype(2.0 ** -32) for w in words], axis=1)  # [num, h, d]
    return np.ascontiguousarray(u)


RANDOM_STREAM_HI = 0x52414E44  # "RAND": high offset word of the random-shooting block streams (icem_sample_piecewise)


def philox_block_uniforms(seed: int, first_block: int, n_blocks: int, d: int, rounds: int = 10, dtype=np.float64):
    """Uniforms ``[n_blocks, d]`` of the device's random-shooting sampler: word 0 of block ``(b, j)``'s stream."""
    return philox_uniforms(seed, RANDOM_STREAM_HI << 32, n_blocks, d, 1, first_index=first_block, rounds=rounds,
                           dtype=dtype)[:, 0, :]


def piecewise_blocks(call_offset: int, count: int, freq: int) -> np.ndarray:
    """Block index of calls ``call_offset .. call_offset+count`` of ``MpcRandom.sample`` (mpc.py:96-102): the action
    drawn at construction serves the first ``freq`` calls, every later draw ``freq + 1`` calls."""
    c = call_offset + np.arange(count, dtype=np.int64)
    return np.where(c < freq, 0, 1 + (c - freq) // (freq + 1))


class RandomShootingOracle:
    """``MpcRandom`` (icem/controllers/mpc.py:86-138): piecewise-constant uniform action sequences, rollout, argmin,
    first action of the best.  ``uniforms(first_block, n_blocks) -> [n_blocks, d]`` supplies the draws."""

    def __init__(self, *, horizon, num_traj, freq, low, high, rollout_cost, uniforms):
        assert freq < horizon  # mpc.py:92
        self.h, self.N, self.freq = horizon, num_traj, freq
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
        self.rollout_cost, self.uniforms = rollout_cost, uniforms
        self.calls = 0

    def sample_action_sequences(self, num_traj: int) -> np.ndarray:
        b = piecewise_blocks(self.calls, num_traj * self.h, self.freq)
        u = np.asarray(self.uniforms(int(b[0]), int(b[-1] - b[0] + 1)), dtype=np.float64)
        self.calls += num_traj * self.h
        acts = (self.high - self.low) * u[b - b[0]] + self.low
        return acts.reshape(num_traj, self.h, -1)

    def get_action(self, obs):
        self.actions = self.sample_action_sequences(self.N)
        self.costs = self.rollout_cost(np.asarray(obs, dtype=np.float64), self.actions)
        self.best = int(np.argmin(self.costs))   # mpc.py:122
        return self.actions[self.best, 0]


class PhiloxNoiseSchedule:
    """Noise callback for :class:`IcemOracle` reproducing the device's Philox
    offsets: per MPC step ``s`` the main batch of iteration ``i`` uses offset
    ``s*(iters+1)+i`` and the shifted-elite batch uses ``s*(iters+1)+iters``.
    The oracle asks for noise in the reference's order (main 0, [shift],
    main 1, ...); the shift batch is recognised by its position."""

    def __init__(self, seed: int, iters: int, d: int, h: int, shift: bool = True,
                 rounds: int = 10, dtype=np.float64, white: bool = False, episode: int = 0):
        self.seed, self.iters, self.d, self.h = seed, iters, d, h
        self.episode = episode  # high word of every offset (the device's icem_set_episode)
        self.shift, self.rounds, self.dtype = shift, rounds, dtype
        self.white = white  # beta <= 0: return (randn[num, h, d], None)
        self.step = -1
        self.begin_step()

    def begin_step(self):
        self.step += 1
        self.it = 0
        self.shift_done = False

    def __call__(self, num: int):
        base = (self.episode << 32) + self.step * (self.iters + 1)
        if self.shift and self.step > 0 and self.it == 1 and not self.shift_done:
            self.shift_done = True
            off = base + self.iters
        else:
            off = base + self.it
            self.it += 1
        if self.white:
            return philox_white_randn(self.seed, off, num, self.d, self.h, 0, self.rounds, self.dtype), None
        return philox_white_noise(self.seed, off, num, self.d, self.h, 0, self.rounds, self.dtype)


# --------------------------------------------------------------------------
# a-2  MpcICem.sample_action_sequences  (icem.py:61-82)
# --------------------------------------------------------------------------


def sample_action_sequences(mean: np.ndarray, std: np.ndarray, low: np.ndarray, high: np.ndarray,
                            beta: float, z_r: np.ndarray, z_i: np.ndarray) -> np.ndarray:
    """``clip(colored[N,h,d]*std + mean, low, high)`` -- icem.py:73-79.
    ``z_r, z_i`` are ``[N, d, F]``; result is ``[N, h, d]``.  ``beta <= 0``: white noise (icem.py:77),
    ``z_r`` is the ``randn(N, h, d)`` draw itself and ``z_i`` is ignored."""
    h = mean.shape[0]
    if beta > 0:
        samples = colored_from_white(beta, h, z_r, z_i).transpose([0, 2, 1])
    else:
        samples = np.asarray(z_r)  # icem.py:77: np.random.randn(num_traj, h, d), passed in as z_r (z_i unused)
    return np.clip(samples * std + mean, low, high)


def sample_via_matrices(mean, std, low, high, beta, z_r, z_i, dtype=np.float64) -> np.ndarray:
    """Same as :func:`sample_action_sequences` through the synthesis matrices
    (the formulation the HIP kernel uses), in ``dtype`` arithmetic, summing
    ``k = 0..F-1`` in order (real term then imaginary term)."""
    h = mean.shape[0]
    Cr, Ci = synthesis_matrices(h, beta)
    Cr = Cr.astype(dtype)
    Ci = Ci.astype(dtype)
    z_r = z_r.astype(dtype)
    z_i = z_i.astype(dtype)
    F = Cr.shape[0]
    y = np.zeros(z_r.shape[:-1] + (h,), dtype=dtype)
    for k in range(F):
        y = y + z_r[..., k:k + 1] * Cr[k]
        y = y + z_i[..., k:k + 1] * Ci[k]
    y = y.transpose([0, 2, 1])
    out = y * std.astype(dtype) + mean.astype(dtype)
    return np.clip(out, low.astype(dtype), high.astype(dtype))


# --------------------------------------------------------------------------
# a-9 / a-10  environment cost functions (mujoco.py:67-99, 259-277)
# --------------------------------------------------------------------------


TERM_NORM, TERM_NORM_GT, TERM_NORM_LT, TERM_SQ_OFFSET, TERM_SUMSQ, TERM_STEP_GT = range(6)


@dataclass(frozen=True)
class CostTerm:
    """``weight * [obs[gate_idx] > gate_thresh] * f`` with ``r = ||obs[a:a+len] - obs[b:b+len]||`` (``b < 0``: the
    slice itself); ``f`` by kind: NORM ``r``, NORM_GT ``[r > thresh]``, NORM_LT ``[r < thresh]``, SQ_OFFSET
    ``(obs[a] - thresh)^2``, SUMSQ ``sum obs[a:a+len]^2``, STEP_GT ``[obs[a] > thresh]``."""
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
    """Parametric restatement of the shipped cost functions.

    ``cost_t = flip_penalty*([obs[flip_idx] > flip_thresh] + [obs[flip_idx] < -flip_thresh])
               + ctrl_weight*sum_d a^2 + lin_weight*obs[lin_idx]
               + diff_weight*(next_obs[diff_idx] - obs[diff_idx])
               + health_penalty*unhealthy(obs)
               + sum_j terms[j]``  (see :class:`CostTerm`)
    added in that order (each group only when switched on: ``flip_idx/diff_idx/health_idx >= 0``;
    the linear term only when ``lin_weight != 0``).  ``unhealthy = 1 - isfinite(obs).all() * [lo <(=) obs[health_idx] <(=) hi]
    * [box_lo < obs[k] < box_hi for all k >= box_from]`` (``health_closed``: ``<=`` as in Ant,
    mujoco.py:146-149; open as in Hopper :189-203 / Humanoid :302-315; the box is Hopper's
    ``healthy_state_range`` over ``obs[2:]``; Hopper's ``healthy_angle`` never enters the result:
    it is passed as ``out=`` of ``np.logical_and``, :199).  The terms cover Reacher (mujoco.py:366-368),
    FetchPickAndPlace / FetchReach dense and sparse (robotics.py:150-164, 286-295), Door and Relocate
    (mjenvs.py:57-78, 155-174).  HalfCheetah (:67-99) and HumanoidStandup (:259-277)
    reproduce the reference bit for bit in float64; the others to rounding (the reference adds its
    terms in a different order per env).
    """
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

    @property
    def needs_next_obs(self) -> bool:
        return self.diff_idx >= 0

    @staticmethod
    def halfcheetah(obs_dim: int = 17, penalise_flipping: bool = True) -> "CostSpec":
        # mujoco.py:77-82: o=18 -> angle [2], velocity [9]; o=17 -> [1], [8]
        if obs_dim == 18:
            a, v = 2, 9
        elif obs_dim == 17:
            a, v = 1, 8
        else:
            raise ValueError(f"Got state of dimension {obs_dim}. Possible dimensions are 17 or 18.")
        return CostSpec(0.1, v, -1.0, a if penalise_flipping else -1, 10.0, math.pi / 2)

    @staticmethod
    def humanoid_standup() -> "CostSpec":
        # mujoco.py:267-272: -obs[2] + 0.1*sum(a^2)
        return CostSpec(0.1, 2, -1.0, -1, 0.0, math.pi / 2)

    @staticmethod
    def ant(dt: float = 0.05, ctrl_cost_weight: float = 0.5, healthy_z_range=(0.2, 1.0)) -> "CostSpec":
        # mujoco.py:151-171 (o=113, positions included): -(next[0]-obs[0])/dt + 100*unhealthy + w*sum(a^2)
        return CostSpec(ctrl_cost_weight, 0, 0.0, -1, 0.0, 0.0, diff_idx=0, diff_weight=-1.0 / dt, health_idx=2,
                        health_penalty=100.0, health_lo=healthy_z_range[0], health_hi=healthy_z_range[1],
                        health_closed=True)

    @staticmethod
    def hopper(dt: float = 0.008, ctrl_cost_weight: float = 1e-3, healthy_z_range=(0.7, float("inf")),
               healthy_state_range=(-100.0, 100.0)) -> "CostSpec":
        # mujoco.py:205-225 (o=12): -(next[0]-obs[0])/dt + 200*unhealthy + w*sum(a^2)
        return CostSpec(ctrl_cost_weight, 0, 0.0, -1, 0.0, 0.0, diff_idx=0, diff_weight=-1.0 / dt, health_idx=1,
                        health_penalty=200.0, health_lo=healthy_z_range[0], health_hi=healthy_z_range[1],
                        box_from=2, box_lo=healthy_state_range[0], box_hi=healthy_state_range[1])

    @staticmethod
    def humanoid(nq: int = 24, exclude_current_positions: bool = True, forward_reward_weight: float = 1.25,
                 ctrl_cost_weight: float = 0.1, healthy_z_range=(1.0, 2.0)) -> "CostSpec":
        # mujoco.py:317-343: -w_f*obs[nq-2 | nq] + 100*unhealthy(z = obs[0 | 2]) + w*sum(a^2)
        return CostSpec(ctrl_cost_weight, nq - 2 if exclude_current_positions else nq, -forward_reward_weight, -1, 0.0,
                        0.0, health_idx=0 if exclude_current_positions else 2, health_penalty=100.0,
                        health_lo=healthy_z_range[0], health_hi=healthy_z_range[1])

    @staticmethod
    def reacher(obs_dim: int = 11) -> "CostSpec":
        # mujoco.py:366-368: ||obs[-3:]||
        return CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=(CostTerm(TERM_NORM, obs_dim - 3, -1, 3),))

    @staticmethod
    def fetch_pick_and_place(orig_obs_len: int = 25, sparse: bool = False, threshold: float = 0.05,
                             shaped_reward: bool = True) -> "CostSpec":
        # robotics.py:150-164: ||goal - obs[3:6]|| (+ 0.1*||obs[0:3] - obs[3:6]||), or the [. > threshold] indicators
        kind = TERM_NORM_GT if sparse else TERM_NORM
        terms = (CostTerm(kind, orig_obs_len, 3, 3, 1.0, threshold),)
        if shaped_reward:
            terms += (CostTerm(kind, 0, 3, 3, 0.1, threshold),)
        return CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms)

    @staticmethod
    def fetch_reach(orig_obs_len: int = 10, sparse: bool = False, threshold: float = 0.05) -> "CostSpec":
        # robotics.py:286-295: ||goal - obs[0:3]|| or [. > threshold]
        return CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0,
                        terms=(CostTerm(TERM_NORM_GT if sparse else TERM_NORM, orig_obs_len, 0, 3, 1.0, threshold),))

    @staticmethod
    def door(obs_dim: int = 39, nq: int = 30, nv: int = 30, shaped_reward: bool = True,
             add_bonus_rewards: bool = True) -> "CostSpec":
        # mjenvs.py:26-31 (index layout), 57-78: 0.1*||palm - handle|| + 0.1*(door - 1.57)^2 + 1e-5*sum(obs[-nv:]^2)
        #                                        - 2[door > 0.2] - 8[door > 1.0] - 10[door > 1.35]
        door, palm, handle = nq - 2, nq - 1, nq + 2
        tail = min(nv, obs_dim)
        terms = ((CostTerm(TERM_NORM, palm, handle, 3, 0.1),) if shaped_reward else ()) + (
            CostTerm(TERM_SQ_OFFSET, door, -1, 1, 0.1, 1.57), CostTerm(TERM_SUMSQ, obs_dim - tail, -1, tail, 1e-5))
        if add_bonus_rewards:
            terms += (CostTerm(TERM_STEP_GT, door, -1, 1, -2.0, 0.2), CostTerm(TERM_STEP_GT, door, -1, 1, -8.0, 1.0),
                      CostTerm(TERM_STEP_GT, door, -1, 1, -10.0, 1.35))
        return CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms)

    @staticmethod
    def relocate(obs_dim: int = 39, nq: int = 36, add_bonus_rewards: bool = True) -> "CostSpec":
        # mjenvs.py:112-115 (index layout), 155-174: 0.1*||palm - obj|| - [obj_z > 0.04]
        #     + 0.5*||obj - target||*[obj_z > 0.04] - 10[||obj - target|| < 0.1] - 20[||obj - target|| < 0.05]
        po, ot, z = nq - 6, nq, obs_dim - 1
        terms = (CostTerm(TERM_NORM, po, -1, 3, 0.1), CostTerm(TERM_STEP_GT, z, -1, 1, -1.0, 0.04),
                 CostTerm(TERM_NORM, ot, -1, 3, 0.5, 0.0, z, 0.04))
        if add_bonus_rewards:
            terms += (CostTerm(TERM_NORM_LT, ot, -1, 3, -10.0, 0.1), CostTerm(TERM_NORM_LT, ot, -1, 3, -20.0, 0.05))
        return CostSpec(0.0, 0, 0.0, -1, 0.0, 0.0, terms=terms)

    def unhealthy(self, obs: np.ndarray) -> np.ndarray:
        z = obs[..., self.health_idx]
        if self.health_closed:
            ok = (self.health_lo <= z) * (z <= self.health_hi)
        else:
            ok = (self.health_lo < z) * (z < self.health_hi)
        if self.box_from >= 0:
            st = obs[..., self.box_from:]
            ok = np.logical_and(np.all(np.logical_and(self.box_lo < st, st < self.box_hi), axis=-1), ok)
        return 1 - np.isfinite(obs).all(axis=-1) * ok

    def term_value(self, tm: CostTerm, obs: np.ndarray) -> np.ndarray:
        """One term; sums of squares accumulated element by element in index order (the order the kernels use)."""
        dt = obs.dtype.type
        if tm.kind == TERM_STEP_GT:
            f = (obs[..., tm.a] > dt(tm.thresh)).astype(obs.dtype)
        elif tm.kind == TERM_SQ_OFFSET:
            v = obs[..., tm.a] - dt(tm.thresh)
            f = v * v
        else:
            acc = np.zeros(obs.shape[:-1], dtype=obs.dtype)
            for m in range(tm.len):
                v = obs[..., tm.a + m]
                if tm.b >= 0:
                    v = v - obs[..., tm.b + m]
                acc = acc + v * v
            if tm.kind == TERM_SUMSQ:
                f = acc
            else:
                r = np.sqrt(acc)
                if tm.kind == TERM_NORM:
                    f = r
                else:
                    f = (r > dt(tm.thresh) if tm.kind == TERM_NORM_GT else r < dt(tm.thresh)).astype(obs.dtype)
        if tm.gate_idx >= 0:
            with np.errstate(invalid="ignore"):   # inf * 0 -> NaN, as in the reference's products
                f = f * (obs[..., tm.gate_idx] > dt(tm.gate_thresh)).astype(obs.dtype)
        return dt(tm.weight) * f

    def __call__(self, obs: np.ndarray, act: np.ndarray, next_obs=None) -> np.ndarray:
        scores = np.zeros(act.shape[:-1], dtype=act.dtype)
        if self.flip_idx >= 0:
            ang = obs[..., self.flip_idx]
            scores = scores + (ang > self.flip_thresh) * self.flip_penalty
            scores = scores + (ang < -self.flip_thresh) * self.flip_penalty
        scores = scores + self.ctrl_weight * np.sum(act ** 2, axis=-1)
        if self.lin_weight != 0:
            scores = scores + self.lin_weight * obs[..., self.lin_idx]
        return scores + self.extended_terms(obs, next_obs)

    def extended_terms(self, obs: np.ndarray, next_obs) -> np.ndarray:
        dt = obs.dtype.type
        ext = np.zeros(obs.shape[:-1], dtype=obs.dtype)
        if self.diff_idx >= 0:
            ext = ext + dt(self.diff_weight) * (next_obs[..., self.diff_idx] - obs[..., self.diff_idx])
        if self.health_idx >= 0:
            ext = ext + dt(self.health_penalty) * self.unhealthy(obs).astype(obs.dtype)
        for tm in self.terms:
            ext = ext + self.term_value(tm, obs)
        return ext


def spec_trajectory_costs(cost: CostSpec, observations, actions, next_observations=None, mode="sum", dtype=None):
    """``trajectory_cost_fn`` (abstract_controller.py:74-91) over rollouts held as arrays
    ``observations / next_observations [P,h,o]``, ``actions [P,h,d]``: per-step costs in the kernels'
    order (control cost summed over d in index order, steps reduced in t order)."""
    dt = np.dtype(actions.dtype if dtype is None else dtype).type
    observations, actions = observations.astype(dt), actions.astype(dt)
    nxt = None if next_observations is None else next_observations.astype(dt)
    acc = None
    for t in range(actions.shape[1]):
        c = _step_cost(cost, observations[:, t], actions[:, t], None if nxt is None else nxt[:, t], dt)
        acc = _reduce_step(acc, c, mode)
    return acc


def _step_cost(cost: CostSpec, obs, a, nxt, dt):
    ctrl = np.zeros(a.shape[0], dtype=dt)
    for j in range(a.shape[1]):
        ctrl = ctrl + a[:, j] * a[:, j]
    c = np.zeros(a.shape[0], dtype=dt)
    if cost.flip_idx >= 0:
        ang = obs[:, cost.flip_idx]
        c = c + (ang > dt(cost.flip_thresh)).astype(dt) * dt(cost.flip_penalty)
        c = c + (ang < dt(-cost.flip_thresh)).astype(dt) * dt(cost.flip_penalty)
    c = c + dt(cost.ctrl_weight) * ctrl
    if cost.lin_weight != 0:
        c = c + dt(cost.lin_weight) * obs[:, cost.lin_idx]
    if cost.extended:
        c = c + cost.extended_terms(obs, nxt)
    return c


def _reduce_step(acc, c, mode):
    if acc is None:
        return c
    if mode == "sum":
        return acc + c
    if mode == "best":
        return np.minimum(acc, c)
    if mode == "final":
        return c
    raise NotImplementedError(mode)


# --------------------------------------------------------------------------
# Synthetic batched forward models (the contract of abstract_models.py:17-53)
# --------------------------------------------------------------------------

MODEL_LINEAR = 0
MODEL_TANH = 1


@dataclass
class SyntheticModel:
    """``o' = act(o @ A + a @ B)`` with ``A [o,o]``, ``B [d,o]``; ``band >= 0``
    keeps only ``|row-col| <= band`` of ``A`` (the masked entries are zeroed
    in ``A`` itself at construction)."""
    A: np.ndarray
    B: np.ndarray
    kind: int = MODEL_LINEAR
    band: int = -1

    def __post_init__(self):
        if self.band >= 0:
            o = self.A.shape[0]
            r = np.arange(o)
            mask = np.abs(r[:, None] - r[None, :]) <= self.band
            self.A = np.where(mask, self.A, 0.0)

    @staticmethod
    def make(o: int, d: int, kind: int = MODEL_LINEAR, band: int = -1,
             seed_a: int = 0, seed_b: int = 1) -> "SyntheticModel":
        # A = 0.95 I + 0.05 N(0,1) (RandomState(0)), B = 0.1 N(0,1) (RandomState(1))
        A = 0.95 * np.eye(o) + 0.05 * np.random.RandomState(seed_a).randn(o, o) / math.sqrt(o)
        B = 0.1 * np.random.RandomState(seed_b).randn(d, o)
        return SyntheticModel(A, B, kind, band)

    def predict(self, obs: np.ndarray, act: np.ndarray) -> np.ndarray:
        # Accumulate in a fixed order (k ascending over obs, then j ascending
        # over actions) so float32 runs are reproducible on the device.
        dt = obs.dtype
        A = self.A.astype(dt)
        B = self.B.astype(dt)
        if dt == np.float64 and A.shape[0] > 64:
            # wide observations (HumanoidStandup's o = 378): one BLAS product per step instead of o passes over
            # [P, o] -- in float64 the summation order is worth 1e-15, and the fixed order matters for float32 only
            nxt = obs @ A + act @ B
            return np.tanh(nxt) if self.kind == MODEL_TANH else nxt
        nxt = np.zeros_like(obs)
        for k in range(A.shape[0]):
            nxt = nxt + obs[..., k:k + 1] * A[k]
        for j in range(B.shape[0]):
            nxt = nxt + act[..., j:j + 1] * B[j]
        if self.kind == MODEL_TANH:
            nxt = np.tanh(nxt)
        return nxt


def rollout_observations(model: SyntheticModel, obs0: np.ndarray, actions: np.ndarray) -> np.ndarray:
    """Batched open-loop rollout (abstract_models.py:17-26): returns the
    *pre-action* observations ``[P, h, o]`` (``observations`` field); the state
    reached by the last action is never scored (mujoco.py cost_fn ignores
    ``next_observations``)."""
    P, h, _ = actions.shape
    obs = np.broadcast_to(obs0.astype(actions.dtype), (P, obs0.shape[0])).copy()
    out = np.empty((P, h, obs0.shape[0]), dtype=actions.dtype)
    for t in range(h):
        out[:, t] = obs
        obs = model.predict(obs, actions[:, t])
    return out


def trajectory_costs(cost_fn: Callable, observations: np.ndarray, actions: np.ndarray,
                     mode: str = "sum") -> np.ndarray:
    """abstract_controller.py:74-91: per-step cost ``[P,h]`` reduced over h."""
    costs_path = cost_fn(observations, actions, None)
    if mode == "sum":
        return np.sum(costs_path, axis=1)
    if mode == "best":
        return np.amin(costs_path, axis=1)
    if mode == "final":
        return costs_path[:, -1]
    raise NotImplementedError("Implement method {} to compute cost along trajectory".format(mode))


def rollout_costs(model: SyntheticModel, cost: CostSpec, obs0, actions, mode="sum", dtype=None):
    """Fused rollout+cost with a running accumulation over ``t`` (the order
    the HIP kernel uses): returns ``costs [P]``."""
    dt = np.dtype(actions.dtype if dtype is None else dtype).type
    actions = actions.astype(dt)
    P, h, _ = actions.shape
    obs = np.broadcast_to(np.asarray(obs0, dtype=dt), (P, len(obs0))).copy()
    acc = None
    for t in range(h):
        a = actions[:, t]
        nxt = model.predict(obs, a)
        acc = _reduce_step(acc, _step_cost(cost, obs, a, nxt, dt), mode)
        obs = nxt
    return acc


def rollout_cost_magnitudes(model: SyntheticModel, cost: CostSpec, obs0, actions):
    """``sum_t sum |addend|`` of every trajectory's "sum"-mode cost (the HalfCheetah / HumanoidStandup form): the scale a
    rounding-error bound on that cost has to be relative to -- a trajectory whose positive and negative terms cancel has
    a small cost but not a small error.  Used by the at-size parity tests (north_star's 1e-5 relative, taken relative to
    this magnitude instead of padded with an absolute floor); same rollout as :func:`rollout_costs`."""
    actions = np.asarray(actions, dtype=np.float64)
    P, h, _ = actions.shape
    obs = np.broadcast_to(np.asarray(obs0, dtype=np.float64), (P, len(obs0))).copy()
    mag = np.zeros(P)
    for t in range(h):
        a = actions[:, t]
        if cost.flip_idx >= 0:
            mag += (np.abs(obs[:, cost.flip_idx]) > cost.flip_thresh) * abs(cost.flip_penalty)
        mag += abs(cost.ctrl_weight) * (a * a).sum(axis=1)
        if cost.lin_weight != 0:
            mag += abs(cost.lin_weight) * np.abs(obs[:, cost.lin_idx])
        for tm in getattr(cost, "terms", ()):   # the term list (icem_cost_terms): |weight x f x gate| of every term
            mag += np.abs(cost.term_value(tm, obs))
        obs = model.predict(obs, a)
    return mag


# --------------------------------------------------------------------------
# a-11  elite selection + refit (icem.py:194-211)
# --------------------------------------------------------------------------


def topk_sorted(costs: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k smallest costs, ascending by ``(cost, index)``; NaN is
    treated as +inf.  The reference's ``argsort()[:k]`` (icem.py:199) is an
    unstable quicksort, identical on tie-free inputs."""
    c = np.where(np.isnan(costs), np.inf, costs)
    order = np.lexsort((np.arange(len(c)), c))
    return order[:k]


def refit(elite_actions: np.ndarray, mean: np.ndarray, std: np.ndarray, alpha: float):
    """icem.py:207-211: mean / population std (ddof=0) over the K elites, then
    momentum ``alpha``."""
    new_mean = elite_actions.mean(axis=0)
    new_std = elite_actions.std(axis=0)
    return (1 - alpha) * new_mean + alpha * mean, (1 - alpha) * new_std + alpha * std


def population_sizes(N: int, K: int, gamma: float, iters: int) -> List[int]:
    """icem.py:123-127: iterated ``max(2K, int(N_{i-1}/gamma))`` (NOT the
    closed form printed at icem.py:38-40)."""
    out, n = [], N
    for i in range(iters):
        if i > 0:
            n = max(K * 2, int(n / gamma))
        out.append(n)
    return out


# --------------------------------------------------------------------------
# a-12 / a-13  the controller loop (icem.py:31-43, 106-189)
# --------------------------------------------------------------------------


@dataclass
class IcemParams:
    horizon: int = 30
    num_simulated_trajectories: int = 128
    factor_decrease_num: float = 1.25
    cost_along_trajectory: str = "sum"
    alpha: float = 0.1
    elites_size: int = 10
    opt_iterations: int = 3
    init_std: float = 0.5
    use_mean_actions: bool = True
    keep_previous_elites: bool = True
    shift_elites_over_time: bool = True
    fraction_elites_reused: float = 0.3
    noise_beta: float = 0.25

    @property
    def num_elites(self) -> int:
        # icem.py:235-240
        return max(2, min(self.elites_size, self.num_simulated_trajectories // 2))


@dataclass
class IterationTrace:
    actions: np.ndarray          # simulated batch [P_sim, h, d]
    costs: np.ndarray            # pool costs [P_pool]
    elite_idx: np.ndarray        # sorted, best first, indices into the pool
    mean: np.ndarray             # after refit
    std: np.ndarray
    best_idx: int


NoiseFn = Callable[[int], Tuple[np.ndarray, np.ndarray]]


class IcemOracle:
    """Array-based restatement of ``MpcICem`` (no Rollout objects).

    ``noise(num_traj)`` must return the white draws ``(z_r, z_i) [num,d,F]``
    (``(randn [num,h,d], None)`` when ``noise_beta <= 0``)
    for one ``sample_action_sequences`` call, in the order the reference makes
    those calls: per MPC step, main batch of iteration 0, then (if elites are
    shifted) the ``(n_reuse, d, h)`` batch, then the main batches of
    iterations 1.. .  ``rollout_cost(obs, actions) -> costs[P]``.
    """

    def __init__(self, params: IcemParams, low: np.ndarray, high: np.ndarray,
                 rollout_cost: Callable[[np.ndarray, np.ndarray], np.ndarray], noise: NoiseFn,
                 compute_new_mean: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None,
                 rollout_obs: Optional[Callable[[np.ndarray, np.ndarray], np.ndarray]] = None):
        # compute_new_mean(last_predicted_obs, kept_last_row) -> new last row: a subclass' override of icem.py:191-192
        # (None: the default, keep the last row); rollout_obs(obs, actions [P,h,d]) -> pre-action observations [P,h,o]
        self.compute_new_mean = compute_new_mean
        self.rollout_obs = rollout_obs
        self.p = params
        self.low = np.asarray(low)
        self.high = np.asarray(high)
        self.d = len(low)
        self.rollout_cost = rollout_cost
        self.noise = noise
        self.was_reset = False
        self.trace: List[List[IterationTrace]] = []

    # icem.py:31-59
    def beginning_of_rollout(self):
        h, d = self.p.horizon, self.d
        self.mean = np.zeros((h, d)) + (self.high + self.low) / 2.0
        self.std = np.ones((h, d)) * (self.high - self.low) / 2.0 * self.p.init_std
        self.elite_actions = None
        self.elite_costs = None
        self.was_reset = True

    def _sample(self, num_traj: int) -> np.ndarray:
        z_r, z_i = self.noise(num_traj)
        return sample_action_sequences(self.mean, self.std, self.low, self.high, self.p.noise_beta, z_r, z_i)

    def get_action(self, obs: np.ndarray) -> np.ndarray:
        if not self.was_reset:
            raise AttributeError("beginning_of_rollout() needs to be called before")
        p = self.p
        K = p.num_elites
        step_trace: List[IterationTrace] = []
        num_sim_traj = p.num_simulated_trajectories
        pool_actions = None
        costs = None
        best = None
        for i in range(p.opt_iterations):
            if i > 0:
                num_sim_traj = max(p.elites_size * 2, int(num_sim_traj / p.factor_decrease_num))
            actions = self._sample(num_sim_traj)                       # icem.py:85
            if p.use_mean_actions and i == p.opt_iterations - 1:        # icem.py:87-88
                actions[0] = self.mean
            if i == 0 and p.shift_elites_over_time and self.elite_actions is not None:
                # icem.py:91-104: best int(K*xi) elites, shifted by one step, new last action
                reused = self.elite_actions[:, 1:]
                n_reuse = int(reused.shape[0] * p.fraction_elites_reused)
                reused = reused[:n_reuse]
                last = self._sample(n_reuse)[:, -1:, :]
                actions = np.concatenate([actions, np.concatenate([reused, last], axis=1)], axis=0)
            sim_costs = self.rollout_cost(obs, actions)                # mpc.py:56-67 + abstract_controller.py:74-91
            pool_actions, costs = actions, sim_costs
            if i > 0 and p.keep_previous_elites:                        # icem.py:143-145
                n_keep = int(len(self.elite_actions) * p.fraction_elites_reused)
                pool_actions = np.concatenate([actions, self.elite_actions[:n_keep]], axis=0)
                costs = np.concatenate([sim_costs, self.elite_costs[:n_keep]])
            best = int(np.argmin(costs))                                # icem.py:149
            idx = topk_sorted(costs, K)                                 # icem.py:199
            self.elite_actions = pool_actions[idx].copy()
            self.elite_costs = costs[idx].copy()
            self.mean, self.std = refit(self.elite_actions, self.mean, self.std, p.alpha)
            step_trace.append(IterationTrace(actions.copy(), costs.copy(), idx.copy(),
                                             self.mean.copy(), self.std.copy(), best))
        executed = pool_actions[best][0].copy()                         # icem.py:163
        self.mean[:-1] = self.mean[1:]                                  # icem.py:167 (last kept: :171,191-192)
        if self.compute_new_mean is not None:                           # icem.py:170-171 under an overriding subclass
            last_predicted_ob = self.rollout_obs(obs, pool_actions[best][None])[0, -1]
            self.mean[-1] = self.compute_new_mean(last_predicted_ob, self.mean[-1].copy())
        self.std = np.ones_like(self.std) * (self.high - self.low) / 2.0 * p.init_std  # icem.py:175
        self.last_min_cost = float(np.min(costs))                       # icem.py:177
        self.trace.append(step_trace)
        return executed


# --------------------------------------------------------------------------
# f-3  MpcCemStd  (icem/controllers/mpc.py:142-327): the CEM baseline with truncated-normal sampling
# --------------------------------------------------------------------------


def truncnorm_from_uniform(u: np.ndarray, lower, upper, mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """``scipy.stats.truncnorm.rvs(lower, upper, loc=mean, scale=std, size=(N, h, d))`` (mpc.py:188-198) given
    the uniform draws ``u [N, h, d]`` scipy takes from the legacy global stream (one ``uniform(size=(N,h,d))`` call,
    verified by the golden generator): the third-party inverse CDF, then the affine map."""
    from scipy.stats import truncnorm
    return truncnorm.ppf(u, lower, upper) * std[None] + mean[None]


def cem_bounds(mean: np.ndarray, std: np.ndarray, low: np.ndarray, high: np.ndarray, like_levine: bool):
    """``MpcCemStd._update_bounds`` (mpc.py:290-301) -> (std, lower, upper); lower / upper broadcast to ``[h, d]``."""
    if like_levine:
        std = np.maximum(1e-8, np.minimum(np.minimum((mean - low) / 2, (high - mean) / 2), std))
        return std, np.full(mean.shape, -2.0), np.full(mean.shape, 2.0)
    return std, (low - mean) / (std + 1e-8), (high - mean) / (std + 1e-8)


class CemStdOracle:
    """Array-based restatement of ``MpcCemStd`` (mpc.py:142-327).  ``uniforms(num) -> u [num, h, d]`` supplies the
    uniform draws of one ``sample_action_sequences`` call; ``rollout_cost(obs, actions) -> costs``."""

    def __init__(self, *, horizon, num_traj, opt_iterations, elites_size, alpha, init_std, like_levine, shift_means,
                 execute_best_elite, low, high, rollout_cost, uniforms):
        self.h, self.N, self.iters = horizon, num_traj, opt_iterations
        self.K = max(2, min(elites_size, num_traj // 2))  # mpc.py:312-316
        self.alpha, self.init_std = alpha, init_std
        self.like_levine, self.shift_means, self.execute_best = like_levine, shift_means, execute_best_elite
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
        self.rollout_cost, self.uniforms = rollout_cost, uniforms
        self.trace = []

    def _init_std(self):
        return np.ones((self.h, len(self.low))) * (self.high - self.low) / 2.0 * self.init_std

    def beginning_of_rollout(self):  # mpc.py:158-170
        self.mean = np.zeros((self.h, len(self.low))) + (self.high + self.low) / 2.0
        self.std, self.lower, self.upper = cem_bounds(self.mean, self._init_std(), self.low, self.high, self.like_levine)

    def get_action(self, obs):  # mpc.py:200-262
        actions = costs = None
        for _ in range(self.iters):
            actions = truncnorm_from_uniform(self.uniforms(self.N), self.lower, self.upper, self.mean, self.std)
            costs = self.rollout_cost(obs, actions)
            idx = topk_sorted(costs, self.K)                        # mpc.py:270
            elites = actions[idx]
            self.mean, self.std = refit(elites, self.mean, self.std, self.alpha)  # mpc.py:277-281
            self.std, self.lower, self.upper = cem_bounds(self.mean, self.std, self.low, self.high, self.like_levine)
            self.trace.append((actions, costs, idx, self.mean.copy(), self.std.copy(), self.lower.copy(), self.upper.copy()))
        best = int(np.argmin(costs))
        executed = actions[best][0].copy() if self.execute_best else self.mean[0].copy()  # mpc.py:230-233
        if self.shift_means:                                          # mpc.py:236-241
            last = self.mean[-1] * 0 if self.like_levine else self.mean[-1].copy()
            self.mean[:-1] = self.mean[1:]
            self.mean[-1] = last
        else:
            self.mean = np.zeros_like(self.mean)
        self.std, self.lower, self.upper = cem_bounds(self.mean, self._init_std(), self.low, self.high, self.like_levine)
        return executed
