"""Build ``libicem_hip.so`` for gfx950 in-tree (``python -m icem_amd.build [--force]``).

One object per translation unit, compiled in parallel; an object is rebuilt when the hash of (compile command, its
source, every header under ``csrc/`` and ``include/icem_hip.h``) changes, the library when any object does.  The hash of
all sources the library was linked from is compiled INTO it (``icem_build_hash()``, a marker string in ``abi.hip``'s
object) so that a stale binary is visible whatever is lying next to it: ``build_info()`` reads the marker straight
from the file (no dlopen) and is what ``bench.py`` prints as ``build``.
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["generic_kernels.hip", "plan.hip", "abi.hip", "exchange.hip", "k_sample.hip", "k_rollout.hip", "k_merge.hip",
         "k_iter_small.hip", "icem_rssm.hip", "icem_rssm_split.hip", "k_rollout_wide.hip", "collective.hip", "k_rollout_ahead.hip",
         "k_rollout_wide_split.hip", "k_rollout_hn.hip", "k_step_xcd.hip"]
OUT = os.path.join(HERE, "libicem_hip.so")
# the same objects with exchange.hip compiled under -DICEM_FAULT_INJECTION (ICEM_XCHG_FAIL drills: tests/test_gpu_exchange_faults.py
# loads it through ICEM_HIP_LIB); the product library carries no fault injection
OUT_FAULTS = os.path.join(HERE, "libicem_hip_faults.so")
FAULT_UNIT = "exchange.hip"
MARK = b"ICEM_BUILD_HASH="  # abi.hip embeds MARK + the 16 hex digits of source_hash()
OBJ = os.path.join(CSRC, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function"]
# per-unit flags (-mllvm -amdgpu-mfma-vgpr-form would keep TileHN's accumulators out of the AGPRs and save its 16-36
# v_accvgpr_read per step)
UNIT_FLAGS = {}   # (measured for TileHN: no gain, 30 more registers)
if os.environ.get("ICEM_DEV_SHAPES"):   # development builds: compile the matrix-pipe kernels for ONE shape, e.g. "30,6,17" (4x faster)
    FLAGS.append(f"-DICEM_FAST_SHAPES(X)=X({os.environ['ICEM_DEV_SHAPES']})")


def _hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _units():
    return [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs + [os.path.join(os.path.dirname(HERE), "include", "icem_hip.h")]


def _digest(paths, extra=""):
    m = hashlib.sha256(extra.encode())
    for p in paths:
        m.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            m.update(f.read())
    return m.hexdigest()[:16]


def source_hash() -> str:
    """Hash of everything the library is built from (sources, headers, flags)."""
    return _digest([os.path.join(CSRC, u) for u in _units()] + _headers(), " ".join(FLAGS) + repr(sorted(UNIT_FLAGS.items())))


def embedded_hash(path: str = OUT):
    """The source hash compiled into a built library (None: no library, or one from before the marker existed)."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(MARK)
    if at < 0:
        return None
    tag = blob[at + len(MARK):at + len(MARK) + 16]
    return tag.decode() if len(tag) == 16 and all(c in b"0123456789abcdef" for c in tag) else None


def build_info() -> dict:
    """{source_hash, built_from, stale}: ``built_from`` is the hash embedded in the library on disk; ``stale`` is True
    when that is not the hash of the sources in the tree."""
    info = {"source_hash": source_hash(), "built_from": embedded_hash(), "stale": True}
    info["stale"] = info["built_from"] != info["source_hash"]
    return info


def up_to_date() -> bool:
    return not build_info()["stale"]


def _compile(unit, headers_digest, verbose, faults=False):
    src = os.path.join(CSRC, unit)
    extra = [f'-DICEM_BUILD_HASH="{source_hash()}"'] if unit == "abi.hip" else []  # the marker lives in one object
    if faults:
        extra.append("-DICEM_FAULT_INJECTION")
    cmd = [_hipcc(), *FLAGS, *UNIT_FLAGS.get(unit, []), *extra, "-I", CSRC, "-c", src]
    key = _digest([src], " ".join(cmd[1:-1]) + headers_digest)
    stem = os.path.splitext(unit)[0] + ("_faults" if faults else "")
    obj = os.path.join(OBJ, f"{stem}.{key}.o")
    if not os.path.exists(obj):
        for f in os.listdir(OBJ):  # drop older objects of this unit
            if f.startswith(stem + ".") and f.endswith(".o"):
                os.remove(os.path.join(OBJ, f))
        if verbose:
            print(" ".join(cmd + ["-o", obj]), flush=True)
        subprocess.check_call(cmd + ["-o", obj + ".tmp"])
        os.replace(obj + ".tmp", obj)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date() and embedded_hash(OUT_FAULTS) == source_hash():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    hd = _digest(_headers())
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        fut_faults = ex.submit(_compile, FAULT_UNIT, hd, verbose, True)
        objs = list(ex.map(lambda u: _compile(u, hd, verbose), _units()))
        obj_faults = fut_faults.result()
    objs_f = [obj_faults if os.path.basename(o).startswith(os.path.splitext(FAULT_UNIT)[0] + ".") else o for o in objs]
    for out, oo in ((OUT, objs), (OUT_FAULTS, objs_f)):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *oo, "-ldl", "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if embedded_hash() != source_hash():
        raise RuntimeError("libicem_hip.so does not carry the hash of the sources it was just built from")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(json.dumps(build_info()))
