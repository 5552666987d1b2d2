"""ctypes loader of the plain-C oracle (oracle/icem_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_I = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libicem_oracle.so")
    src = os.path.join(_HERE, "icem_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    lib = C.CDLL(so)
    lib.icem_c_num_threads.restype = C.c_int
    lib.icem_c_set_threads.argtypes = [C.c_int]
    lib.icem_c_noise_tables.argtypes = [C.c_int, C.c_double, _D, _D]
    lib.icem_c_sample_clip.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_int64,
                                       C.c_int, _D, _D, _D, _D, C.c_int, _D]
    lib.icem_c_rollout_cost.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _D, _D,
                                        C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, _D]
    lib.icem_c_topk.argtypes = [C.c_int, C.c_int, _D, _I, _D]
    lib.icem_c_refit.argtypes = [C.c_int, C.c_int, C.c_double, _D, _I, _D, _D]
    lib.icem_c_iteration.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64,
                                     C.c_uint64, C.c_int, C.c_int, _D, _D, _D, _D, _D, C.c_double, C.c_int,
                                     C.c_double, C.c_int, C.c_double, C.c_double, _D, _D, _D, _D, _I, _D]
    _LIB = lib
    return lib


def c64(x):
    return np.ascontiguousarray(x, dtype=np.float64)
