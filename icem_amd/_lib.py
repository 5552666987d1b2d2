"""ctypes binding of ``libicem_hip.so`` (the C ABI declared in ``include/icem_hip.h``)."""
from __future__ import annotations

import ctypes as C
import os

# dmabuf IPC (hipIpcGetMemHandle of the exchange blocks, RCCL): must be in the environment before the HSA runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

ICEM_F32, ICEM_F64 = 0, 1
COST_MODES = {"sum": 0, "best": 1, "final": 2}
MODEL_LINEAR, MODEL_TANH = 0, 1
(BUF_MEAN, BUF_STD, BUF_LOW, BUF_HIGH, BUF_OBS0, BUF_ACTIONS, BUF_COSTS, BUF_ELITES, BUF_RECORDS,
 BUF_WORKSPACE, BUF_EXECUTED, BUF_BEST_COST, BUF_COUNT) = range(13)

KERNEL_NAMES = ["sample_clip", "rollout_cost", "topk_partial", "local_pack", "merge_refit", "sample_rollout"]

ICEM_E_INVALID, ICEM_E_UNSUPPORTED, ICEM_E_HIP, ICEM_E_NO_DEVICE, ICEM_E_STATE, ICEM_E_RANGE = -1, -2, -3, -4, -5, -6
ERR_NAMES = {-1: "ICEM_E_INVALID", -2: "ICEM_E_UNSUPPORTED", -3: "ICEM_E_HIP", -4: "ICEM_E_NO_DEVICE",
             -5: "ICEM_E_STATE", -6: "ICEM_E_RANGE"}


class IcemError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class IcemConfigC(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32), ("act_dim", C.c_int32), ("num_traj", C.c_int32), ("num_elites", C.c_int32),
        ("elites_size", C.c_int32), ("opt_iters", C.c_int32), ("cost_mode", C.c_int32),
        ("use_mean_actions", C.c_int32), ("keep_previous_elites", C.c_int32), ("shift_elites", C.c_int32),
        ("dtype", C.c_int32), ("rng_rounds", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
        ("factor_decrease", C.c_double), ("alpha", C.c_double), ("init_std", C.c_double),
        ("fraction_reused", C.c_double), ("noise_beta", C.c_double), ("seed", C.c_uint64),
    ]


class IcemCostSpecC(C.Structure):
    _fields_ = [("ctrl_weight", C.c_double), ("lin_weight", C.c_double), ("flip_penalty", C.c_double),
                ("flip_thresh", C.c_double), ("lin_idx", C.c_int32), ("flip_idx", C.c_int32)]


MAX_COST_TERMS = 8
TERM_NORM, TERM_NORM_GT, TERM_NORM_LT, TERM_SQ_OFFSET, TERM_SUMSQ, TERM_STEP_GT = range(6)


class IcemCostTermC(C.Structure):
    """include/icem_hip.h: struct icem_cost_term."""
    _fields_ = [("weight", C.c_double), ("thresh", C.c_double), ("gate_thresh", C.c_double), ("kind", C.c_int32),
                ("a", C.c_int32), ("b", C.c_int32), ("len", C.c_int32), ("gate_idx", C.c_int32), ("reserved", C.c_int32)]


class IcemCostTermsC(C.Structure):
    """include/icem_hip.h: struct icem_cost_terms."""
    _fields_ = [("diff_weight", C.c_double), ("health_penalty", C.c_double), ("health_lo", C.c_double),
                ("health_hi", C.c_double), ("box_lo", C.c_double), ("box_hi", C.c_double),
                ("diff_idx", C.c_int32), ("health_idx", C.c_int32), ("health_closed", C.c_int32), ("box_from", C.c_int32),
                ("n_terms", C.c_int32), ("reserved", C.c_int32), ("terms", IcemCostTermC * MAX_COST_TERMS)]


class IcemPlanBuffersC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "mean", "std", "low", "high", "obs0", "actions", "costs", "elites", "records", "workspace",
        "executed", "best_cost", "z_r", "z_i", "z_r_shift", "z_i_shift")]


# every symbol include/icem_hip.h declares: (name, restype, argtypes)
_VP, _I32, _I64, _U64, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
_H = C.c_void_p
SYMBOLS = [
    ("icem_abi_version", C.c_int, []),
    ("icem_build_hash", C.c_char_p, []),
    ("icem_last_error", C.c_char_p, []),
    ("icem_device_count", C.c_int, []),
    ("icem_create", C.c_int, [C.POINTER(IcemConfigC), C.POINTER(_H)]),
    ("icem_destroy", C.c_int, [_H]),
    ("icem_population_sizes", C.c_int, [_H, C.POINTER(_I32)]),
    ("icem_noise_tables_host", C.c_int, [_I32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("icem_set_model", C.c_int, [_H, _I32, _I32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("icem_set_cost", C.c_int, [_H, C.POINTER(IcemCostSpecC)]),
    ("icem_rssm_param_elems", _SZ, []),
    ("icem_rssm_rollout_cost", C.c_int, [_I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP]),
    ("icem_get_action", C.c_int, [_H, C.POINTER(IcemPlanBuffersC), _I32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), _VP]),
    ("icem_set_cost_terms", C.c_int, [_H, C.POINTER(IcemCostTermsC)]),
    ("icem_trajectory_cost", C.c_int, [_H, _I32, _I32, _VP, _VP, _I64, _I64, _VP, _VP, _VP]),
    ("icem_sample_clip", C.c_int, [_H, _I32, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _U64, _I32, _I32, _VP, _VP]),
    ("icem_sample_piecewise", C.c_int, [_H, _I32, _I64, _I32, _I64, _VP, _VP, _VP, _VP, _VP]),
    ("icem_philox_normals", C.c_int, [_H, _I32, _I64, _U64, _VP, _VP, _VP]),
    ("icem_rollout_cost", C.c_int, [_H, _I32, _VP, _VP, _VP, _VP, _VP]),
    ("icem_cost_reduce", C.c_int, [_H, _I32, _VP, _VP, _VP]),
    ("icem_topk_workspace_bytes", _SZ, [_H, _I32, _I32]),
    ("icem_topk_sorted", C.c_int, [_H, _I32, _VP, _I32, _VP, _VP, _VP, _VP]),
    ("icem_gather_refit", C.c_int, [_H, _VP, _VP, _I32, _VP, _VP, _VP, _VP]),
    ("icem_update_distribution", C.c_int, [_H, _I32, _VP, _VP, _I32, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("icem_shift", C.c_int, [_H, _VP, _VP, _VP, _VP, _VP]),
    ("icem_reset_distribution", C.c_int, [_H, _VP, _VP, _VP, _VP, _VP]),
    ("icem_plan_buffer_bytes", _SZ, [_H, _I32]),
    ("icem_plan_iter_local", C.c_int, [_H, C.POINTER(IcemPlanBuffersC), _I32, _I32, _VP]),
    ("icem_plan_iter_merge", C.c_int, [_H, C.POINTER(IcemPlanBuffersC), _I32, _I32, _VP]),
    ("icem_plan_step", C.c_int, [_H, C.POINTER(IcemPlanBuffersC), _I32, _VP]),
    ("icem_record_bytes", _SZ, [_H]),
    ("icem_profile_enable", C.c_int, [_H, _I32]),
    ("icem_debug_stamps", C.c_int, [_H, _VP]),
    ("icem_set_merge_deferral", C.c_int, [_H, C.c_int32]),
    ("icem_set_episode", C.c_int, [_H, C.c_uint64]),
    ("icem_exchange_create", C.c_int, [_H, _VP]),
    ("icem_exchange_connect", C.c_int, [_H, _VP, C.POINTER(C.c_void_p)]),
    ("icem_exchange_block", C.c_void_p, [_H]),
    ("icem_exchange_disable", C.c_int, [_H]),
    ("icem_exchange_status", C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("icem_plan_step_sharded", C.c_int, [_H, C.POINTER(IcemPlanBuffersC), _I32, _VP]),
    ("icem_exchange_probe", C.c_int, [_H, C.c_int32, _VP, C.POINTER(C.c_double)]),
    ("icem_sample_truncnorm", C.c_int, [_H, C.c_int32, C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_uint64, _VP, _VP]),
    ("icem_cem_bounds", C.c_int, [_H, C.c_int32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("icem_update_distribution_ok", C.c_int, [_H, C.c_int32, C.c_int32]),
    ("icem_rccl_load", C.c_int, [C.c_char_p]),
    ("icem_rccl_library", C.c_char_p, []),
    ("icem_rccl_unique_id", C.c_int, [_VP]),
    ("icem_rccl_connect", C.c_int, [_H, _VP]),
    ("icem_rccl_adopt", C.c_int, [_H, _VP]),
    ("icem_rccl_disconnect", C.c_int, [_H]),
    ("icem_allgather_elites", C.c_int, [_H, _VP, _VP]),
    ("icem_profile_read", C.c_int, [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("icem_rssm_trim", C.c_int, []),
    ("icem_set_wide_exact", C.c_int, [_H, _I32]),
    ("icem_set_wide_arith", C.c_int, [_H, _I32]),
    ("icem_wide_arith", C.c_int, [_H]),
    ("icem_wide_imbalance_log2", C.c_int, [_H]),
    ("icem_wide_model_imbalance_log2", C.c_int, [_I32, _I32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("icem_set_tile_arith", C.c_int, [_H, _I32]),
    ("icem_tile_arith", C.c_int, [_H]),
    ("icem_profile_overhead", C.c_int, [_VP, _I32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("icem_plan_step_batch", C.c_int, [C.POINTER(_H), _I32, C.POINTER(IcemPlanBuffersC), _I32, _VP]),
    ("icem_batch_uploads", C.c_int64, [_H]),
    ("icem_step_status", C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _VP]),
    ("icem_tile_growth", C.c_double, [_H]),
    ("icem_nonfinite_costs", C.c_int, [_H, C.POINTER(C.c_int64), _VP]),
    ("icem_set_option", C.c_int, [C.c_char_p, C.c_double]),
    ("icem_get_option", C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    ("icem_reset_options", C.c_int, []),
    ("icem_option_name", C.c_char_p, [_I32]),
]


IPC_HANDLE_BYTES = 64
RCCL_ID_BYTES = 128
ABI_VERSION = 5   # include/icem_hip.h: ICEM_ABI_VERSION


def lib_path() -> str:
    return os.environ.get("ICEM_HIP_LIB", os.path.join(_HERE, "libicem_hip.so"))


def load_library() -> C.CDLL:
    """Load ``libicem_hip.so`` and bind every exported symbol.  Raises if the
    library has not been built -- there is deliberately no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'"
                          f" or python -m icem_amd.build)")
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.icem_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.icem_abi_version()} != {ABI_VERSION}: rebuild (python -m icem_amd.build)")
    _LIB = lib
    return lib


# ---- development options (include/icem_hip.h: icem_set_option; icem_amd/csrc/options.h) ----------------------------------
def option_names():
    lib = load_library()
    out, i = [], 0
    while True:
        n = lib.icem_option_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def set_option(name: str, value: float):
    check(load_library().icem_set_option(name.encode(), float(value)))


def get_option(name: str) -> float:
    v = C.c_double()
    check(load_library().icem_get_option(name.encode(), C.byref(v)))
    return v.value


def reset_options():
    check(load_library().icem_reset_options())


# spellings the environment variables of rounds 3-5 used for options that are numbers now
_ENV_WORDS = {"thread": 0.0, "rows": 1.0}


def apply_env_options(environ=None) -> dict:
    """Map ``ICEM_<NAME>`` environment variables onto the library's options -- for TOOLS (bench.py, tools/, test child
    processes), called explicitly: neither the library nor ``load_library()`` reads the environment.  Returns what was
    set.  (``ICEM_GK_ROLLOUT=thread`` is option ``gk_rollout_thread`` = 1, ``ICEM_GK_SAMPLE=thread`` is ``gk_sample`` = 0.)"""
    environ = os.environ if environ is None else environ
    done = {}
    for name in option_names():
        key = "ICEM_" + name.upper()
        if key in environ and environ[key] != "":
            raw = environ[key]
            val = _ENV_WORDS[raw] if raw in _ENV_WORDS else float(raw)
            set_option(name, val)
            done[name] = val
    if environ.get("ICEM_GK_ROLLOUT", "") == "thread":
        set_option("gk_rollout_thread", 1.0)
        done["gk_rollout_thread"] = 1.0
    return done


_FOLLOW_ENV = False


def follow_environment(on: bool = True):
    """TOOLS ONLY (tools/, the soaks): from now on every new ``IcemPlanner`` / learned-dynamics model first resets the
    options and applies the ``ICEM_<NAME>`` variables of the moment (:func:`apply_env_options`) -- the scripts of rounds 3-5
    flip ``os.environ`` between planners.  Off by default: product code never looks at the environment."""
    global _FOLLOW_ENV
    _FOLLOW_ENV = bool(on)
    if on:
        reset_options()
        apply_env_options()


def maybe_follow_environment():
    if _FOLLOW_ENV:
        reset_options()
        apply_env_options()


def check(rc: int):
    if rc != 0:
        msg = load_library().icem_last_error()
        raise IcemError(rc, msg.decode() if msg else "")
