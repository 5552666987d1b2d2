"""Build ``libicem_hip.so`` for gfx950 in-tree (``python -m icem_amd.build``)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", "icem_kernels.hip"), os.path.join(HERE, "csrc", "icem_fused.hip"),
        os.path.join(HERE, "csrc", "icem_rssm.hip")]
OUT = os.path.join(HERE, "libicem_hip.so")
DEPS = SRCS + [os.path.join(HERE, "csrc", "philox.h"), os.path.join(HERE, "csrc", "icem_fused.h"), os.path.join(HERE, "csrc", "icem_rssm.h"),
               os.path.join(os.path.dirname(HERE), "include", "icem_hip.h")]


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-parallel-jobs=4", *SRCS, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
