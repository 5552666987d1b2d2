"""The one-wave selections of the merge prologues against a host sort, on crafted candidate lists (tests/units/select_equiv.hip,
compiled here with hipcc against the kernels' own header): merge_select<12> with the kept elites inserted / offered apart,
merge_select_stream, and merge_select_shallow<3> (the noise-ahead launch's: EXPERIMENTS R6.17) -- random lists, the K best
clustered in one or three lists (survivors deeper than the registers hold), ties at the threshold below and above 64 survivors,
fewer than K finite keys, kept elites that win or lose, launches with fewer than 64 lists."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_wave_selections_agree_with_a_host_sort(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "select_equiv")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-I", os.path.join(ROOT, "icem_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "units", "select_equiv.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 mismatching" in out.stdout, out.stdout + out.stderr
