"""Torch-tensor level wrapper of the C ABI: one :class:`IcemPlanner` per controller.

PyTorch is plumbing here (device memory, streams, ``torch.distributed``); all
arithmetic happens in ``libicem_hip.so``.  Each method names the reference
call site it replaces (paths relative to ``/root/reference``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .distributed import exchange_records, shard_range


@dataclass
class IcemConfig:
    """Constructor kwargs of the reference's ``MpcICem`` (icem/controllers/icem.py:213-233,
    icem/controllers/mpc.py:22) plus the device-side knobs."""
    horizon: int
    act_dim: int
    num_traj: int
    elites_size: int = 10
    opt_iters: int = 3
    cost_mode: str = "sum"
    use_mean_actions: bool = True
    keep_previous_elites: bool = True
    shift_elites: bool = True
    factor_decrease: float = 1.25
    alpha: float = 0.1
    init_std: float = 0.5
    fraction_reused: float = 0.3
    noise_beta: float = 0.25
    dtype: str = "f32"
    rng_rounds: int = 10
    seed: int = 0
    rank: int = 0
    world: int = 1

    @property
    def num_elites(self) -> int:
        # icem/controllers/icem.py:235-240
        return max(2, min(self.elites_size, self.num_traj // 2))

    @property
    def torch_dtype(self):
        return torch.float64 if self.dtype == "f64" else torch.float32

    def to_c(self) -> L.IcemConfigC:
        if self.cost_mode not in L.COST_MODES:
            raise NotImplementedError(
                "Implement method {} to compute cost along trajectory".format(self.cost_mode))
        return L.IcemConfigC(
            horizon=self.horizon, act_dim=self.act_dim, num_traj=self.num_traj, num_elites=self.num_elites,
            elites_size=self.elites_size, opt_iters=self.opt_iters, cost_mode=L.COST_MODES[self.cost_mode],
            use_mean_actions=int(self.use_mean_actions), keep_previous_elites=int(self.keep_previous_elites),
            shift_elites=int(self.shift_elites), dtype=L.ICEM_F64 if self.dtype == "f64" else L.ICEM_F32,
            rng_rounds=self.rng_rounds, rank=self.rank, world=self.world, factor_decrease=self.factor_decrease,
            alpha=self.alpha, init_std=self.init_std, fraction_reused=self.fraction_reused,
            noise_beta=self.noise_beta, seed=self.seed & 0xFFFFFFFFFFFFFFFF)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# noise callback of the parity mode: noise(num_traj) -> (z_r, z_i) float64 arrays [num, d, F]
# (noise_beta <= 0, the white branch of icem.py:77: -> (randn [num, h, d], None))
NoiseFn = Callable[[int], Tuple[np.ndarray, np.ndarray]]


class IcemPlanner:
    """Owns one ``icem_handle`` and the device buffers of one controller."""

    def __init__(self, cfg: IcemConfig, low, high, device="cuda:0", process_group=None):
        self.lib = L.load_library()
        L.maybe_follow_environment()   # (tools only: icem_amd._lib.follow_environment)
        if not torch.cuda.is_available():
            raise RuntimeError("icem_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.dt = cfg.torch_dtype
        self.group = process_group
        torch.cuda.set_device(self.device)
        self._h = C.c_void_p()
        ccfg = cfg.to_c()
        L.check(self.lib.icem_create(C.byref(ccfg), C.byref(self._h)))
        if L._FOLLOW_ENV and os.environ.get("ICEM_TILE_ARITH", "") != "":   # (tools only; the arithmetic is the HANDLE's)
            L.check(self.lib.icem_set_tile_arith(self._h, int(os.environ["ICEM_TILE_ARITH"])))
        self.h, self.d, self.K = cfg.horizon, cfg.act_dim, cfg.num_elites
        self.F = self.h // 2 + 1
        self.low = torch.as_tensor(np.asarray(low, dtype=np.float64), dtype=self.dt, device=self.device).contiguous()
        self.high = torch.as_tensor(np.asarray(high, dtype=np.float64), dtype=self.dt, device=self.device).contiguous()
        if self.low.shape != (self.d,) or self.high.shape != (self.d,):
            raise ValueError("low/high must have shape [act_dim]")
        pops = (C.c_int32 * cfg.opt_iters)()
        L.check(self.lib.icem_population_sizes(self._h, pops))
        self.population_sizes = list(pops)
        self.n_reuse = int(self.K * cfg.fraction_reused)
        self.obs_dim = 0
        self._bufs = None
        self.mpc_step = 0
        self.episode = 0           # folded into the device noise streams (icem_set_episode)
        self._episodes_started = 0
        self._topk_ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.icem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ model / cost
    def set_model(self, kind: int, A: np.ndarray, B: np.ndarray):
        """Built-in batched model ``o' = act(o@A + a@B)`` (predict contract of
        icem/models/abstract_models.py:17-26)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        o = A.shape[0]
        if A.shape != (o, o) or B.shape != (self.d, o):
            raise ValueError("A must be [o,o] and B [act_dim,o]")
        L.check(self.lib.icem_set_model(self._h, kind, o, A.ctypes.data_as(C.POINTER(C.c_double)),
                                        B.ctypes.data_as(C.POINTER(C.c_double))))
        self.obs_dim = o
        self._bufs = None

    def set_cost(self, ctrl_weight=0.1, lin_idx=8, lin_weight=-1.0, flip_idx=1, flip_penalty=10.0,
                 flip_thresh=float(np.pi / 2)):
        """Parametric HalfCheetah / HumanoidStandup cost (icem/environments/mujoco.py:67-99, 259-277)."""
        spec = L.IcemCostSpecC(ctrl_weight, lin_weight, flip_penalty, flip_thresh, lin_idx, flip_idx)
        L.check(self.lib.icem_set_cost(self._h, C.byref(spec)))

    def sample_piecewise(self, n: int, call_offset: int, change_freq: int, u=None, first_block: int = 0) -> torch.Tensor:
        """``MpcRandom.sample_action_sequences`` (icem/controllers/mpc.py:96-109): ``[n, h, d]`` uniform actions held
        over consecutive ``sample()`` calls (``icem_sample_piecewise``).  ``u``: the draws ``[*, d]`` of blocks
        ``first_block..`` (parity), or None for the device's Philox streams."""
        actions = torch.empty((n, self.h, self.d), dtype=self.dt, device=self.device)
        u_t = None if u is None else self._t(u)
        L.check(self.lib.icem_sample_piecewise(self._h, n, int(call_offset), int(change_freq), int(first_block),
                                               _ptr(self.low), _ptr(self.high), _ptr(u_t), _ptr(actions), self._stream()))
        return actions

    def set_cost_spec(self, spec):
        """The whole parametric cost of an env (``envs.CostSpec``): the HalfCheetah / HumanoidStandup form plus the
        Ant / Hopper / Humanoid / Reacher / Fetch terms (``icem_set_cost_terms``)."""
        self.set_cost(spec.ctrl_weight, spec.lin_idx, spec.lin_weight, spec.flip_idx, spec.flip_penalty, spec.flip_thresh)
        if not getattr(spec, "extended", False):
            L.check(self.lib.icem_set_cost_terms(self._h, None))
            return
        if len(spec.terms) > L.MAX_COST_TERMS:
            raise ValueError(f"at most {L.MAX_COST_TERMS} cost terms")
        t = L.IcemCostTermsC(
            diff_weight=spec.diff_weight, health_penalty=spec.health_penalty, health_lo=spec.health_lo,
            health_hi=spec.health_hi, box_lo=spec.box_lo, box_hi=spec.box_hi, diff_idx=spec.diff_idx,
            health_idx=spec.health_idx, health_closed=int(spec.health_closed), box_from=spec.box_from,
            n_terms=len(spec.terms))
        for j, tm in enumerate(spec.terms):
            t.terms[j] = L.IcemCostTermC(weight=tm.weight, thresh=tm.thresh, gate_thresh=tm.gate_thresh, kind=tm.kind,
                                         a=tm.a, b=tm.b, len=tm.len, gate_idx=tm.gate_idx)
        L.check(self.lib.icem_set_cost_terms(self._h, C.byref(t)))

    def trajectory_cost(self, observations, actions, next_observations=None) -> torch.Tensor:
        """``trajectory_cost_fn`` (abstract_controller.py:74-91) on the device for rollouts held as tensors:
        ``observations`` / ``next_observations`` ``[n, h, o]`` (any strides over n and h, entries of a row
        contiguous -- a transposed view of a step-major ``[h, n, o]`` buffer works), ``actions [n, h, d]``."""
        n, h, o = observations.shape
        if h != self.h or observations.dtype != self.dt or observations.stride(2) != 1:
            raise ValueError("observations must be [n, horizon, o] of the planner dtype with contiguous rows")
        if next_observations is not None and (next_observations.shape != observations.shape
                                              or next_observations.stride() != observations.stride()
                                              or next_observations.dtype != self.dt):
            raise ValueError("next_observations must match observations in shape, strides and dtype")
        actions = self._t(actions, (n, self.h, self.d))
        costs = torch.empty((n,), dtype=self.dt, device=self.device)
        L.check(self.lib.icem_trajectory_cost(
            self._h, n, o, _ptr(observations), _ptr(next_observations) if next_observations is not None else None,
            observations.stride(0), observations.stride(1), _ptr(actions), _ptr(costs), self._stream()))
        return costs

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _t(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=self.dt, device=self.device).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def noise_tables(self) -> Tuple[np.ndarray, np.ndarray]:
        cr = np.zeros((self.F, self.h))
        ci = np.zeros((self.F, self.h))
        L.check(self.lib.icem_noise_tables_host(self.h, self.cfg.noise_beta, cr.ctypes.data_as(C.POINTER(C.c_double)),
                                                ci.ctypes.data_as(C.POINTER(C.c_double))))
        return cr, ci

    # ------------------------------------------------------------------ stateless operators
    def sample_clip(self, n: int, mean, std, z_r=None, z_i=None, offset: int = 0, first_index: int = 0,
                    t_begin: int = 0, row0_mean: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """K1: ``MpcICem.sample_action_sequences`` (icem/controllers/icem.py:61-82)."""
        mean = self._t(mean, (self.h, self.d))
        std = self._t(std, (self.h, self.d))
        if z_r is not None and self.cfg.noise_beta <= 0:  # white branch (icem.py:77): the randn(n, h, d) draw itself
            z_r, z_i = self._t(z_r, (n, self.h, self.d)), None
        elif z_r is not None:
            z_r = self._t(z_r, (n, self.d, self.F))
            z_i = self._t(z_i, (n, self.d, self.F))
        if out is None:
            out = torch.empty((n, self.h, self.d), dtype=self.dt, device=self.device)
        L.check(self.lib.icem_sample_clip(self._h, n, first_index, _ptr(mean), _ptr(std), _ptr(self.low),
                                          _ptr(self.high), _ptr(z_r), _ptr(z_i), offset, t_begin, int(row0_mean),
                                          _ptr(out), self._stream()))
        return out

    def philox_normals(self, n: int, offset: int = 0, first_index: int = 0):
        z_r = torch.empty((n, self.d, self.F), dtype=self.dt, device=self.device)
        z_i = torch.empty_like(z_r)
        L.check(self.lib.icem_philox_normals(self._h, n, first_index, offset, _ptr(z_r), _ptr(z_i), self._stream()))
        return z_r, z_i

    def rollout_cost(self, obs0, actions: torch.Tensor, return_observations: bool = False):
        """K2: ``simulate_trajectories`` + ``trajectory_cost_fn`` (icem/controllers/mpc.py:56-67,
        icem/controllers/abstract_controller.py:74-91) with the built-in model."""
        obs0 = self._t(obs0, (self.obs_dim,))
        actions = self._t(actions)
        n = actions.shape[0]
        if tuple(actions.shape[1:]) != (self.h, self.d):
            raise ValueError("actions must be [n, h, d]")
        costs = torch.empty((n,), dtype=self.dt, device=self.device)
        obs = torch.empty((n, self.h, self.obs_dim), dtype=self.dt, device=self.device) if return_observations else None
        L.check(self.lib.icem_rollout_cost(self._h, n, _ptr(obs0), _ptr(actions), _ptr(costs), _ptr(obs), self._stream()))
        return (costs, obs) if return_observations else costs

    def cost_reduce(self, step_costs) -> torch.Tensor:
        step_costs = self._t(step_costs)
        n = step_costs.shape[0]
        if step_costs.shape[1] != self.h:
            raise ValueError("step_costs must be [n, h]")
        costs = torch.empty((n,), dtype=self.dt, device=self.device)
        L.check(self.lib.icem_cost_reduce(self._h, n, _ptr(step_costs), _ptr(costs), self._stream()))
        return costs

    def topk_sorted(self, costs, k: Optional[int] = None):
        """K3: ``np.array(costs).argsort()[:K]`` (icem/controllers/icem.py:199)."""
        costs = self._t(costs)
        n = costs.shape[0]
        k = self.K if k is None else k
        nbytes = self.lib.icem_topk_workspace_bytes(self._h, n, k)
        if self._topk_ws is None or self._topk_ws.numel() < nbytes:
            self._topk_ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=self.device)
        out_c = torch.empty((k,), dtype=self.dt, device=self.device)
        out_i = torch.empty((k,), dtype=torch.int32, device=self.device)
        L.check(self.lib.icem_topk_sorted(self._h, n, _ptr(costs), k, _ptr(out_c), _ptr(out_i), _ptr(self._topk_ws),
                                          self._stream()))
        return out_c, out_i

    def gather_refit(self, actions: torch.Tensor, idx: torch.Tensor, mean: torch.Tensor, std: torch.Tensor):
        """K4: ``update_distributions`` (icem/controllers/icem.py:201-211); mean/std updated in place."""
        assert mean.dtype == self.dt and std.dtype == self.dt and mean.is_contiguous() and std.is_contiguous()
        actions = self._t(actions)
        idx = idx.to(device=self.device, dtype=torch.int32).contiguous()
        k = idx.shape[0]
        elites = torch.empty((k, self.h, self.d), dtype=self.dt, device=self.device)
        L.check(self.lib.icem_gather_refit(self._h, _ptr(actions), _ptr(idx), k, _ptr(mean), _ptr(std), _ptr(elites),
                                           self._stream()))
        return elites

    def can_update_in_one_launch(self, n_all: int, k: int) -> bool:
        """Does this handle serve ``icem_update_distribution`` for ``n_all`` candidates and ``k`` elites?  (f32, pools of at
        most 16 384 candidates, k <= 32 -- and the fast-path switch the HANDLE latched at ``icem_create``: asked of the
        handle, not re-read from the environment.)"""
        return bool(self.lib.icem_update_distribution_ok(self._h, int(n_all), int(k)))

    def update_distribution(self, costs: torch.Tensor, pool: torch.Tensor, k: int, mean: torch.Tensor, std: torch.Tensor,
                            keep_costs: Optional[torch.Tensor] = None, keep_actions: Optional[torch.Tensor] = None):
        """``update_distributions`` (icem/controllers/icem.py:194-211) with the kept elites appended behind the pool
        (icem.py:143-145) in ONE launch: -> (elite costs [k], indices [k] into [pool | kept], elites [k, h, d]); ``mean`` /
        ``std`` refitted in place.  Same bits as ``topk_sorted`` over the concatenation + ``gather_refit``."""
        assert mean.dtype == self.dt and std.dtype == self.dt and mean.is_contiguous() and std.is_contiguous()
        n = costs.shape[0]
        costs, pool = self._t(costs, (n,)), self._t(pool, (n, self.h, self.d))
        n_keep = 0 if keep_costs is None else keep_costs.shape[0]
        if n_keep:
            keep_costs, keep_actions = self._t(keep_costs, (n_keep,)), self._t(keep_actions, (n_keep, self.h, self.d))
        elites = torch.empty((k, self.h, self.d), dtype=self.dt, device=self.device)
        ec = torch.empty(k, dtype=self.dt, device=self.device)
        idx = torch.empty(k, dtype=torch.int32, device=self.device)
        L.check(self.lib.icem_update_distribution(self._h, n, _ptr(costs), _ptr(pool), n_keep,
                                                  _ptr(keep_costs) if n_keep else None, _ptr(keep_actions) if n_keep else None,
                                                  k, _ptr(mean), _ptr(std), _ptr(elites), _ptr(ec), _ptr(idx), self._stream()))
        return ec, idx, elites

    def sample_truncnorm(self, n: int, mean, std, lower, upper, u=None, offset: int = 0, first_index: int = 0,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``MpcCemStd.sample_action_sequences`` (icem/controllers/mpc.py:188-198): truncated-normal samples
        ``[n, h, d]``; ``lower`` / ``upper`` are ``[h, d]`` in standard-normal units; ``u`` = scipy's uniform draws
        ``[n, h, d]`` (parity mode) or None (device RNG)."""
        mean, std = self._t(mean, (self.h, self.d)), self._t(std, (self.h, self.d))
        lower, upper = self._t(lower, (self.h, self.d)), self._t(upper, (self.h, self.d))
        if u is not None:
            u = self._t(u, (n, self.h, self.d))
        if out is None:
            out = torch.empty((n, self.h, self.d), dtype=self.dt, device=self.device)
        L.check(self.lib.icem_sample_truncnorm(self._h, n, first_index, _ptr(mean), _ptr(std), _ptr(lower), _ptr(upper),
                                               _ptr(u), offset, _ptr(out), self._stream()))
        return out

    def cem_bounds(self, mean: torch.Tensor, std: torch.Tensor, like_levine: bool):
        """``MpcCemStd._update_bounds`` (icem/controllers/mpc.py:290-301): returns (lower, upper) ``[h, d]``;
        ``std`` is capped in place when ``like_levine``."""
        assert mean.dtype == self.dt and std.dtype == self.dt and mean.is_contiguous() and std.is_contiguous()
        lower = torch.empty((self.h, self.d), dtype=self.dt, device=self.device)
        upper = torch.empty_like(lower)
        L.check(self.lib.icem_cem_bounds(self._h, int(like_levine), _ptr(mean), _ptr(std), _ptr(self.low), _ptr(self.high),
                                         _ptr(lower), _ptr(upper), self._stream()))
        return lower, upper

    def shift(self, mean: torch.Tensor, std: torch.Tensor):
        """Epilogue of get_action (icem/controllers/icem.py:167-175), in place."""
        L.check(self.lib.icem_shift(self._h, _ptr(mean), _ptr(std), _ptr(self.low), _ptr(self.high), self._stream()))

    def reset_distribution(self, mean: torch.Tensor, std: torch.Tensor):
        """beginning_of_rollout (icem/controllers/icem.py:31-59), in place."""
        L.check(self.lib.icem_reset_distribution(self._h, _ptr(mean), _ptr(std), _ptr(self.low), _ptr(self.high),
                                                 self._stream()))

    def set_tile_arith(self, mode="auto"):
        """16 <= padded obs_dim <= 20: the arithmetic of the tile kernels' model step (``icem_set_tile_arith``): ``"f32"`` / 0
        the exact fmaf chain, ``"f16x2"`` / 1 two fp16 planes per operand on the 16-bit matrix cores, ``"auto"`` / -1 by the
        configuration's global population.  Returns the arithmetic now in effect (0 / 1)."""
        m = {"auto": -1, "f32": 0, "f16x2": 1}.get(mode, mode)
        L.check(self.lib.icem_set_tile_arith(self._h, int(m)))
        return self.tile_arith

    @property
    def tile_arith(self):
        return int(self.lib.icem_tile_arith(self._h))

    @property
    def tile_growth(self) -> float:
        """Reachable maximum of |state entry| / max(|obs0|, action bound) over the horizon for the handle's model
        (``icem_tile_growth``): what decides whether the fp16-plane tiles are served (<= 2^10)."""
        return float(self.lib.icem_tile_growth(self._h))

    def step_status(self) -> Tuple[int, bool]:
        """(MPC steps served by the one-launch kernel of small populations, whether one of its bounded waits ever ran out)
        -- ``icem_step_status``; synchronises the launch stream."""
        n, t = C.c_int64(), C.c_int32()
        L.check(self.lib.icem_step_status(self._h, C.byref(n), C.byref(t), self._stream()))
        return int(n.value), bool(t.value)

    def nonfinite_costs(self) -> int:
        """Trajectories since the planner was made whose cost left a tile kernel NaN (``icem_nonfinite_costs``;
        synchronises the launch stream)."""
        n = C.c_int64()
        L.check(self.lib.icem_nonfinite_costs(self._h, C.byref(n), self._stream()))
        return int(n.value)

    WIDE_ARITH = {"auto": -1, "f16x2": 0, "fp16x2": 0, "f32": 1, "bf16x3": 2}

    def set_wide_arith(self, mode="auto"):
        """obs_dim > 32: which arithmetic the model step's GEMM runs in (``icem_set_wide_arith``), by NAME: ``"f32"`` the
        exact-f32 matrix pipe (bitwise an fmaf chain -- strict parity), ``"f16x2"`` two fp16 planes per operand (three
        products per multiply-add), ``"bf16x3"`` three bf16 planes (six products, operands exact at any magnitude),
        ``"auto"`` (the library's default) f16x2 unless the balanced model keeps a row or column more than 2^13 below its
        largest weight -- then bf16x3.  Returns the arithmetic now in effect (a name)."""
        m = self.WIDE_ARITH.get(mode, mode)
        L.check(self.lib.icem_set_wide_arith(self._h, int(m)))
        return self.wide_arith

    @property
    def wide_arith(self):
        """The wide arithmetic in effect for the current model: ``"f16x2"`` / ``"f32"`` / ``"bf16x3"``."""
        return {0: "f16x2", 1: "f32", 2: "bf16x3"}[int(self.lib.icem_wide_arith(self._h))]

    @property
    def wide_imbalance_log2(self):
        return int(self.lib.icem_wide_imbalance_log2(self._h))

    def set_wide_exact(self, on=True):
        """ABI <= 3 spelling of :meth:`set_wide_arith`: ``False`` / 0 the fp16 planes, ``True`` / 1 exact f32, 2 the bf16
        planes (``on`` defaults to exact f32; the LIBRARY's default is ``"auto"``)."""
        L.check(self.lib.icem_set_wide_exact(self._h, int(on)))

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, on: bool = True):
        L.check(self.lib.icem_profile_enable(self._h, int(on)))

    def profile_read(self):
        """{kernel: (total_ms, launches, units)} since the last read (HIP events on the launch stream)."""
        n = len(L.KERNEL_NAMES)
        ms, cnt, units = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
        L.check(self.lib.icem_profile_read(self._h, ms, cnt, units))
        return {L.KERNEL_NAMES[i]: (ms[i], cnt[i], units[i]) for i in range(n) if cnt[i]}

    def profile_overhead(self, reps: int = 200, spin_us: float = 15.0) -> Tuple[float, float]:
        """(event-pair time around a one-wave kernel spinning for ``spin_us``, how long that kernel really ran) in
        microseconds, on the launch stream (``icem_profile_overhead``): their difference is the bracket's share of every
        ``profile_read`` span."""
        pair, kern = C.c_double(), C.c_double()
        L.check(self.lib.icem_profile_overhead(self._stream(), reps, spin_us, C.byref(pair), C.byref(kern)))
        return pair.value, kern.value

    def get_action_host(self, obs) -> Tuple[np.ndarray, float]:
        """One MPC step for a host caller (``icem_get_action``): float64 observation in, ``(executed action [d] float64,
        best cost of the last pool)`` out; one H2D copy, the step's launches, one D2H copy and one synchronisation, all
        inside the library.  Device noise, world == 1."""
        self._ensure_buffers()
        self._cb.z_r = self._cb.z_i = self._cb.z_r_shift = self._cb.z_i_shift = None
        ob = np.ascontiguousarray(obs, dtype=np.float64)
        if ob.shape != (self.obs_dim,):
            raise ValueError(f"expected an observation of shape ({self.obs_dim},)")
        act = np.empty(self.d, dtype=np.float64)
        best = C.c_double()
        rc = self.lib.icem_get_action(self._h, C.byref(self._cb), self.mpc_step, ob.ctypes.data_as(C.POINTER(C.c_double)),
                                      act.ctypes.data_as(C.POINTER(C.c_double)), C.byref(best), self._stream())
        if rc in (0, L.ICEM_E_RANGE):
            self.mpc_step += 1   # (ICEM_E_RANGE: the step ran and its outputs are filled -- but they are not the reference's)
        L.check(rc)
        return act, best.value

    def plan_step_resident(self):
        """plan_step with the observation already in ``self.obs0`` (Philox noise): no host work besides
        the launches (and, for world > 1, the one all-gather per iteration) -- the bench's timed region."""
        self._cb.z_r = self._cb.z_i = self._cb.z_r_shift = self._cb.z_i_shift = None
        st = self._stream()
        if self.cfg.world == 1:
            L.check(self.lib.icem_plan_step(self._h, C.byref(self._cb), self.mpc_step, st))
        elif getattr(self, "_exchange", False) or getattr(self, "_rccl", False):
            # in-library exchange (or, failing that, the library's own RCCL all-gather on the launch stream): one C call
            # per MPC step, no host-side collective
            L.check(self.lib.icem_plan_step_sharded(self._h, C.byref(self._cb), self.mpc_step, st))
        else:
            # non-last merges ride in the next iteration's launch (nobody looks at mean / std in between)
            self._set_deferral(True)
            gather = self._resident_gather()
            local, merge, cb, h, step = self.lib.icem_plan_iter_local, self.lib.icem_plan_iter_merge, C.byref(self._cb), self._h, self.mpc_step
            for it in range(self.cfg.opt_iters):
                L.check(local(h, cb, step, it, st))
                gather()
                L.check(merge(h, cb, step, it, st))
        self.mpc_step += 1

    # ------------------------------------------------------------------ B independent planners, one launch per stage
    @staticmethod
    def plan_step_batch(planners: Sequence["IcemPlanner"], observations=None):
        """One MPC step of every planner of ``planners`` (one configuration; models, costs, seeds and observations of their
        own) as ``icem_plan_step_batch``: the reference's parallel episodes (icem/misc/rollout_utils.py:46-58, 129-152),
        every stage one launch for all of them.  ``observations``: one per planner (``None``: already in ``planner.obs0``).
        Each planner's buffers afterwards are bit for bit those of its own :meth:`plan_step`.  Returns the executed actions
        (device tensors, no host sync)."""
        pls = list(planners)
        n = len(pls)
        if n == 0:
            return []
        lib = pls[0].lib
        step = pls[0].mpc_step
        for i, pl in enumerate(pls):
            pl._ensure_buffers()
            if pl.mpc_step != step:
                raise ValueError("the planners of a batch advance together: same mpc_step")
            pl._cb.z_r = pl._cb.z_i = pl._cb.z_r_shift = pl._cb.z_i_shift = None
            if observations is not None:
                pl.obs0.copy_(torch.as_tensor(np.asarray(observations[i], dtype=np.float64), dtype=pl.dt), non_blocking=False)
        hs = (C.c_void_p * n)(*[pl._h for pl in pls])
        bs = (L.IcemPlanBuffersC * n)(*[pl._cb for pl in pls])
        L.check(lib.icem_plan_step_batch(hs, n, bs, step, pls[0]._stream()))
        for pl in pls:
            pl.mpc_step += 1
        return [pl.executed for pl in pls]

    @property
    def batch_uploads(self) -> int:
        return int(self.lib.icem_batch_uploads(self._h))

    # ------------------------------------------------------------------ in-library elite exchange (world > 1)
    def connect_exchange(self, group=None):
        """Set up the in-library elite exchange between the ranks' processes (``icem_exchange_create`` /
        ``icem_exchange_connect``): every rank allocates its exchange block, the 64-byte IPC handles travel ONCE over
        ``torch.distributed`` (any backend), and from then on an MPC step makes no host-side collective: the records
        move as peer-to-peer stores issued by the library's own kernels (xGMI between GPUs)."""
        import torch.distributed as dist
        self._ensure_buffers()
        group = self.group if group is None else group
        mine = (C.c_ubyte * L.IPC_HANDLE_BYTES)()
        err = None
        try:
            L.check(self.lib.icem_exchange_create(self._h, mine))
        except L.IcemError as e:
            err = e
        handles = [None] * self.cfg.world
        dist.all_gather_object(handles, None if err else bytes(mine), group=group)
        if err is None and all(hd is not None for hd in handles):
            blob = (C.c_ubyte * (L.IPC_HANDLE_BYTES * self.cfg.world)).from_buffer_copy(b"".join(handles))
            try:
                L.check(self.lib.icem_exchange_connect(self._h, blob, None))
            except L.IcemError as e:
                err = e
        elif err is None:
            err = L.IcemError(L.ICEM_E_HIP if hasattr(L, "ICEM_E_HIP") else -3, "a peer could not create its exchange block")
        # all ranks or none: one rank that cannot map its peers sends everybody back to the host-driven all-gather
        oks = [None] * self.cfg.world
        dist.all_gather_object(oks, err is None, group=group)
        if all(oks):
            # ... and it has to WORK, not only map: a few real exchanges (every rank pushes to every rank and waits for
            # all of them); a rank whose waits time out (peer stores that never land) sends everybody back as well
            self._exchange = True
            try:
                self.exchange_probe(rounds=4)
                ok = self.exchange_status()[0] == 0
            except L.IcemError as e:
                ok, err = False, e
            dist.all_gather_object(oks, ok, group=group)
            if all(oks):
                return True
            if err is None:
                err = L.IcemError(-3, "the exchange self-test timed out on a rank")
        self.lib.icem_exchange_disable(self._h)
        self._exchange = False
        self.exchange_error = str(err) if err is not None else "a peer failed to connect"
        return False

    def degrade_exchange(self, group=None) -> str:
        """COLLECTIVE, after a run-time failure of the records' path (a bounded wait for a peer ran out: the next
        ``icem_plan_step_sharded`` raises): every rank leaves the path it was on and takes the next one in the order in-library
        exchange -> the library's RCCL all-gather -> host-driven all-gather; the distribution is re-initialised (the steps
        since the failure planned on garbage).  Returns the path now in use: ``"rccl"`` or ``"host"``."""
        torch.cuda.synchronize(self.device)
        was_exchange = bool(getattr(self, "_exchange", False))
        if was_exchange:
            status = self.exchange_status()[0] | int(getattr(self, "_xchg_status_seen", 0))   # read and clear
            self._xchg_status_seen = 0
            why = getattr(self, "_degrade_reason", None)   # (set by the caller when the trigger was not a timeout)
            self._degrade_reason = None
            self.exchange_error = (f"run time: {why} (status word {status})" if why and not (status & 1) else
                                   f"run time: a wait for a peer's elite records timed out (status word {status})")
            L.check(self.lib.icem_exchange_disable(self._h))
            self._exchange = False
        elif getattr(self, "_rccl", False):
            self.rccl_error = "run time: the in-library all-gather failed"
            self.lib.icem_rccl_disconnect(self._h)
            self._rccl = False
        self._gather_fn = None
        self.reset()
        if was_exchange and self.connect_rccl(group):
            return "rccl"
        return "host"

    def connect_rccl(self, group=None) -> bool:
        """The fallback of :meth:`connect_exchange`: an RCCL communicator owned by the library (``icem_rccl_connect``) so
        that ``icem_plan_step_sharded`` gathers the ranks' records with ``ncclAllGather`` on the launch stream
        (``icem_allgather_elites``) -- still one C call per MPC step and no host-side collective.  The 128-byte
        ``ncclUniqueId`` travels once over ``torch.distributed`` (any backend).  All ranks or none."""
        import torch.distributed as dist
        self._ensure_buffers()
        group = self.group if group is None else group
        err = None
        ident = [None]
        # RCCL wants one GPU per rank (two ranks on one device: "duplicate GPU", or a hang inside ncclCommInitRank):
        # find out BEFORE anybody enters the blocking call
        import socket
        where = [None] * self.cfg.world
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        mine = (socket.gethostname(), dev_index,
                tuple(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")))
        dist.all_gather_object(where, mine, group=group)
        if len(set(where)) < self.cfg.world:
            self._rccl = False
            self.rccl_error = "two ranks share a GPU: RCCL needs one device per rank"
            return False
        try:
            # bind the copy of RCCL this process already carries (torch's), else torch's file, else the system's
            # (ICEM_RCCL_LIB names another one: read HERE, in the tooling -- the library reads no environment variable)
            L.check(self.lib.icem_rccl_load((os.environ.get("ICEM_RCCL_LIB") or
                                             os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")).encode()))
            if self.cfg.rank == 0:
                buf = (C.c_ubyte * L.RCCL_ID_BYTES)()
                L.check(self.lib.icem_rccl_unique_id(buf))
                ident = [bytes(buf)]
        except L.IcemError as e:
            err = e
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        oks = [None] * self.cfg.world
        dist.all_gather_object(oks, err is None and ident[0] is not None, group=group)
        if all(oks):   # (ncclCommInitRank blocks until every rank has called it: only enter it together)
            try:
                blob = (C.c_ubyte * L.RCCL_ID_BYTES).from_buffer_copy(ident[0])
                with torch.cuda.device(self.device):   # ncclCommInitRank binds the CURRENT device: the planner's, not torch's default
                    L.check(self.lib.icem_rccl_connect(self._h, blob))
                    # ... and one real all-gather, checked: every rank's K record slots marked with its number
                    K, r = self.K, self.cfg.rank
                    self.records.zero_()
                    self.records[r * K:(r + 1) * K] = float(r + 1)
                    L.check(self.lib.icem_allgather_elites(self._h, _ptr(self.records), self._stream()))
                    torch.cuda.synchronize(self.device)
                    want = torch.arange(1, self.cfg.world + 1, device=self.device, dtype=self.dt).repeat_interleave(K)
                    if not bool((self.records == want[:, None]).all().item()):
                        raise RuntimeError("the in-library all-gather returned other ranks' records in the wrong slots or not at all")
                    self.records.zero_()
            except (L.IcemError, RuntimeError) as e:
                err = e
            dist.all_gather_object(oks, err is None, group=group)
            if all(oks):
                self._rccl = True
                return True
        self.lib.icem_rccl_disconnect(self._h)
        self._rccl = False
        self.rccl_error = str(err) if err is not None else "a peer failed to create the communicator"
        return False

    @staticmethod
    def connect_exchange_local(planners: Sequence["IcemPlanner"]):
        """The same for ranks that live in ONE process (tests, several GPUs driven by one host thread): the blocks are
        handed over as device pointers, no IPC."""
        for pl in planners:
            pl._ensure_buffers()
            scratch = (C.c_ubyte * L.IPC_HANDLE_BYTES)()
            L.check(pl.lib.icem_exchange_create(pl._h, scratch))
        blocks = (C.c_void_p * len(planners))(*[pl.lib.icem_exchange_block(pl._h) for pl in planners])
        for pl in planners:
            L.check(pl.lib.icem_exchange_connect(pl._h, None, blocks))
            pl._exchange = True

    def exchange_status(self) -> Tuple[int, bool]:
        """(status, finegrained): status != 0 means a device-side wait for a peer timed out since the last call."""
        s, f = C.c_int32(), C.c_int32()
        L.check(self.lib.icem_exchange_status(self._h, C.byref(s), C.byref(f)))
        return s.value, bool(f.value)

    def exchange_probe(self, rounds: int = 200) -> float:
        """Average latency [us] of one in-library exchange (collective: all ranks call it together)."""
        us = C.c_double()
        L.check(self.lib.icem_exchange_probe(self._h, rounds, self._stream(), C.byref(us)))
        return us.value

    def _resident_gather(self):
        """The per-iteration all-gather of the bench loop with everything that does not change hoisted out: on RCCL
        it is one in-place ``all_gather_into_tensor`` of this rank's K records."""
        g = getattr(self, "_gather_fn", None)
        if g is None:
            import torch.distributed as dist
            rank, world, K, records, group = self.cfg.rank, self.cfg.world, self.K, self.records, self.group
            if records.is_cuda and dist.get_backend(group) == "nccl":
                mine = records[rank * K:(rank + 1) * K]
                g = lambda: dist.all_gather_into_tensor(records, mine, group=group)  # noqa: E731
            else:
                g = lambda: exchange_records(records, K, rank, world, group)  # noqa: E731
            self._gather_fn = g
        return g

    def _set_deferral(self, on: bool):
        if getattr(self, "_deferral", False) != on:
            L.check(self.lib.icem_set_merge_deferral(self._h, int(on)))
            self._deferral = on

    # ------------------------------------------------------------------ fused MPC step
    def _ensure_buffers(self):
        if self._bufs is not None:
            return
        if self.obs_dim == 0:
            raise RuntimeError("set_model()/set_cost() must be called before planning")
        names = ["mean", "std", "low", "high", "obs0", "actions", "costs", "elites", "records", "workspace",
                 "executed", "best_cost"]
        t = {}
        for which, name in enumerate(names):
            nbytes = self.lib.icem_plan_buffer_bytes(self._h, which)
            if name in ("low", "high"):
                t[name] = self.low if name == "low" else self.high
                continue
            t[name] = torch.zeros((max(1, nbytes),), dtype=torch.uint8, device=self.device)
        self._bufs = t
        self.mean = t["mean"].view(self.dt).view(self.h, self.d)
        self.std = t["std"].view(self.dt).view(self.h, self.d)
        self.obs0 = t["obs0"].view(self.dt)
        self.executed = t["executed"].view(self.dt)
        self.best_cost = t["best_cost"].view(self.dt)
        self.actions = t["actions"].view(self.dt).view(-1, self.h, self.d)
        self.costs = t["costs"].view(self.dt)
        rs = self.h * self.d + 2
        self.records = t["records"].view(self.dt).view(self.cfg.world * self.K, rs)
        el = t["elites"].view(self.dt)
        khd = self.K * self.h * self.d
        self.elites_actions = el[:2 * khd].view(2, self.K, self.h, self.d)
        self.elites_costs = el[2 * khd:].view(2, self.K)
        self._cb = L.IcemPlanBuffersC(**{n: t[n].data_ptr() for n in names})
        self._cb.z_r = self._cb.z_i = self._cb.z_r_shift = self._cb.z_i_shift = None

    def reset(self):
        """``MpcICem.beginning_of_rollout`` (icem/controllers/icem.py:31-43)."""
        self._ensure_buffers()
        self.reset_distribution(self.mean, self.std)
        self.mpc_step = 0

    def new_episode(self) -> int:
        """Move the device noise streams on to the next episode (``icem_set_episode``): the reference's ``np.random``
        stream runs on across episodes (icem/controllers/icem.py:73-77), so rollouts must not replay each other's
        exploration noise.  The first call selects episode 0.  Returns the episode number."""
        self.episode = self._episodes_started
        self._episodes_started += 1
        L.check(self.lib.icem_set_episode(self._h, self.episode))
        return self.episode

    def noise_offset(self, call: int) -> int:
        """Stream offset of sampling call ``call`` of the current episode (what ``icem_plan_*`` use internally)."""
        return (self.episode << 32) + int(call)

    def current_elites(self):
        """Elite actions [K,h,d] and costs [K] of the latest iteration (best first)."""
        g = (self.mpc_step * self.cfg.opt_iters) & 1
        return self.elites_actions[g], self.elites_costs[g]

    def local_count(self, it: int) -> int:
        lo, hi = shard_range(self.population_sizes[it], self.cfg.rank, self.cfg.world)
        return hi - lo

    def plan_step(self, obs, noise: Optional[NoiseFn] = None, on_iteration=None) -> torch.Tensor:
        """One MPC step = the loop of ``MpcICem.get_action`` (icem/controllers/icem.py:123-175).
        Returns the executed action as a device tensor ``[d]`` (no host sync).  ``noise`` switches to
        the parity mode: the white draws come from the callback (in the reference's call order)."""
        self._ensure_buffers()
        self.obs0.copy_(torch.as_tensor(np.asarray(obs, dtype=np.float64), dtype=self.dt), non_blocking=False)
        cfg = self.cfg
        st = self._stream()
        xchg = getattr(self, "_exchange", False)
        if noise is None and on_iteration is None and (cfg.world == 1 or xchg or getattr(self, "_rccl", False)):
            self._cb.z_r = self._cb.z_i = self._cb.z_r_shift = self._cb.z_i_shift = None
            L.check(self.lib.icem_plan_step_sharded(self._h, C.byref(self._cb), self.mpc_step, st))
        else:
            self._set_deferral(False)  # callers of this form look at the buffers between iterations
            keep = []
            for it in range(cfg.opt_iters):
                self._cb.z_r = self._cb.z_i = self._cb.z_r_shift = self._cb.z_i_shift = None
                if noise is not None:
                    n_it = self.population_sizes[it]
                    lo, hi = shard_range(n_it, cfg.rank, cfg.world)
                    z_r, z_i = noise(n_it)  # the reference draws the whole batch (icem.py:73 / :77)
                    zr = self._t(z_r[lo:hi])
                    zi = self._t(z_i[lo:hi]) if z_i is not None and cfg.noise_beta > 0 else None
                    keep += [zr, zi]
                    self._cb.z_r, self._cb.z_i = zr.data_ptr(), (zi.data_ptr() if zi is not None else None)
                    if it == 0 and cfg.shift_elites and self.mpc_step > 0 and self.n_reuse > 0:
                        s_r, s_i = noise(self.n_reuse)  # icem.py:102
                        sr = self._t(s_r)
                        si = self._t(s_i) if s_i is not None and cfg.noise_beta > 0 else None
                        keep += [sr, si]
                        self._cb.z_r_shift = sr.data_ptr()
                        self._cb.z_i_shift = si.data_ptr() if si is not None else None
                L.check(self.lib.icem_plan_iter_local(self._h, C.byref(self._cb), self.mpc_step, it, st))
                if cfg.world > 1 and not xchg:  # (connected exchange: the local call has pushed the records already)
                    exchange_records(self.records, self.K, cfg.rank, cfg.world, self.group)
                L.check(self.lib.icem_plan_iter_merge(self._h, C.byref(self._cb), self.mpc_step, it, st))
                if on_iteration is not None:
                    on_iteration(it)
            if keep:
                torch.cuda.current_stream(self.device).synchronize()  # z tensors must outlive the kernels
        self.mpc_step += 1
        return self.executed
