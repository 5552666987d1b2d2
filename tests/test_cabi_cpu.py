"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/icem_hip.h
declares, and its host-only helpers agree with the oracle.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from icem_amd import _lib as L
from oracle import icem_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.lib_path()):
        import __graft_entry__ as g
        g.build()
    return L.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "icem_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(icem_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in L.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)


def test_abi_version_and_error_string(lib):
    assert lib.icem_abi_version() == L.ABI_VERSION == 5
    assert isinstance(lib.icem_last_error(), bytes)


def test_build_hash_is_compiled_into_the_library(lib):
    """A stale binary must be visible whatever stamp files lie around: the hash of the sources is a string inside the
    .so (icem_build_hash), read by icem_amd.build.build_info() without loading it."""
    from icem_amd import build as B
    info = B.build_info()
    assert info["built_from"] == info["source_hash"] and not info["stale"]
    assert lib.icem_build_hash().decode() == info["source_hash"]
    assert B.embedded_hash(os.path.join(ROOT, "include", "icem_hip.h")) is None   # a file without the marker


def test_struct_sizes_match_header():
    # icem_config: 14 int32 + 5 double + uint64 = 56 + 48 = 104 bytes, no padding surprises
    assert C.sizeof(L.IcemConfigC) == 104
    assert C.sizeof(L.IcemCostSpecC) == 40
    assert C.sizeof(L.IcemPlanBuffersC) == 16 * 8
    # icem_cost_term: 3 double + 6 int32 = 48; icem_cost_terms: 6 double + 6 int32 + 8 terms = 48 + 24 + 384 = 456
    # (checked against `gcc sizeof` of include/icem_hip.h below when a C compiler is around)
    assert C.sizeof(L.IcemCostTermC) == 48 and C.sizeof(L.IcemCostTermsC) == 456


def test_struct_sizes_match_c_compiler(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "icem_hip.h")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu", sizeof(icem_config), '
                   'sizeof(icem_cost_spec), sizeof(icem_plan_buffers), sizeof(icem_cost_term), sizeof(icem_cost_terms));return 0;}' % hdr)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(x) for x in (L.IcemConfigC, L.IcemCostSpecC, L.IcemPlanBuffersC, L.IcemCostTermC, L.IcemCostTermsC)]


def test_create_rejects_bad_config_like_the_reference(lib):
    from icem_amd.planner import IcemConfig
    h = C.c_void_p()
    cfg = IcemConfig(horizon=30, act_dim=6, num_traj=1).to_c()
    rc = lib.icem_create(C.byref(cfg), C.byref(h))
    assert rc == -1 and b"At least two trajectories needed!" in lib.icem_last_error()  # mpc.py:30-31
    cfg = IcemConfig(horizon=30, act_dim=6, num_traj=64, noise_beta=float("nan")).to_c()
    assert lib.icem_create(C.byref(cfg), C.byref(h)) == -1  # (beta <= 0 is valid: the white branch, icem.py:77)
    with pytest.raises(NotImplementedError):
        IcemConfig(horizon=30, act_dim=6, num_traj=64, cost_mode="median").to_c()


@pytest.mark.parametrize("h,beta", [(30, 0.25), (12, 0.25), (13, 1.0), (30, 2.0), (10, 3.0), (64, 0.5), (2, 1.0), (3, 1.0)])
def test_noise_tables_match_oracle(lib, h, beta):
    F = h // 2 + 1
    cr = np.zeros((F, h))
    ci = np.zeros((F, h))
    assert lib.icem_noise_tables_host(h, beta, cr.ctypes.data_as(C.POINTER(C.c_double)),
                                      ci.ctypes.data_as(C.POINTER(C.c_double))) == 0
    Cr, Ci = O.synthesis_matrices(h, beta)
    np.testing.assert_allclose(cr, Cr, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(ci, Ci, rtol=1e-13, atol=1e-15)
    # and through them, the reference's irfft formulation
    rs = np.random.RandomState(0)
    zr, zi = rs.randn(5, 3, F), rs.randn(5, 3, F)
    np.testing.assert_allclose(zr @ cr + zi @ ci, O.colored_from_white(beta, h, zr, zi), rtol=0, atol=1e-13)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from icem_amd import IcemConfig, IcemPlanner
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        IcemPlanner(IcemConfig(horizon=30, act_dim=6, num_traj=64), -np.ones(6), np.ones(6))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "icem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "icem_oracle" not in src, f


def test_rssm_parameter_packing_matches_the_library_layout(lib):
    """The Python packer of the fused learned-dynamics kernel (icem_amd.models.pack_rssm) and the layout compiled into
    the library (icem_amd/csrc/icem_rssm.h) agree on the buffer size; the packed blocks hold the weights where the
    kernel's lane indexing expects them (block (ob, kb), lane 16*g + i, slot v  <-  W[16*ob + i][32*kb + 8*g + v])."""
    import torch
    from icem_amd.models import declared_rssm, pack_rssm
    m = declared_rssm(seed=1, device="cpu")
    packed = pack_rssm(m.module)
    assert packed.dtype == torch.int16 and packed.numel() == lib.icem_rssm_param_elems()
    # first layer (inp: [200, 36] -> padded [208, 64], z at columns 0..29, a at 32..37): spot-check three entries
    W = m.module.inp.weight.detach()
    blk = packed[:13 * 2 * 512].view(13, 2, 4, 16, 8)          # [ob, kb, g, i, v]
    as_bf16 = lambda x: x.to(torch.bfloat16).view(torch.int16)  # noqa: E731
    assert blk[0, 0, 0, 0, 0] == as_bf16(W[0, 0])               # z column 0 of output 0
    assert blk[3, 0, 2, 5, 7] == as_bf16(W[3 * 16 + 5, 2 * 8 + 7])
    assert blk[12, 1, 0, 7, 3] == as_bf16(W[12 * 16 + 7, 30 + 3])   # action column 3 sits at padded column 32 + 3
    assert int(blk[12, 1, 3].abs().sum()) == 0                  # padded action columns 56..63 are zero


def test_rccl_binding_is_lazy_and_reports(lib):
    """collective.hip binds RCCL at run time: loading libicem_hip.so must not need librccl, icem_allgather_elites without a
    handle is an argument error, and where an RCCL is present a unique id can be made without a GPU."""
    import subprocess
    out = subprocess.run(["readelf", "-d", L.lib_path()], capture_output=True, text=True).stdout
    assert "rccl" not in out.lower()
    assert lib.icem_allgather_elites(None, None, None) == L.ICEM_E_INVALID
    ident = (C.c_ubyte * L.RCCL_ID_BYTES)()
    rc = lib.icem_rccl_unique_id(ident)
    assert rc in (0, L.ICEM_E_UNSUPPORTED, L.ICEM_E_HIP), lib.icem_last_error()
    if rc == 0:
        assert any(bytes(ident)) and b"rccl" in lib.icem_rccl_library()


def test_quoted_numbers_follow_from_the_tracked_profiles():
    """Every figure DESIGN.md section 5 and profiles/README.md quote for the round is generated from the files under
    profiles/ (tools/doc_numbers.py): the committed documents must be what the script generates, not hand copies."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "doc_numbers.py"), "r06", "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_wide_auto_criterion_on_host_models(lib):
    """ICEM_WIDE_AUTO's measure (icem_wide_model_imbalance_log2: host code, no device): small for the benchmark's model and
    a dense Gaussian one, an entry's worth of decades for a block in other units and for a model whose rows' weights are
    mostly far below their largest; an all-but-dead row alone does not trip it (the rows are equilibrated)."""
    import ctypes as C
    import math

    def imb(A, B):
        A, B = np.ascontiguousarray(A, np.float64), np.ascontiguousarray(B, np.float64)
        return lib.icem_wide_model_imbalance_log2(A.shape[0], B.shape[0], A.ctypes.data_as(C.POINTER(C.c_double)),
                                                  B.ctypes.data_as(C.POINTER(C.c_double)))
    for o, d in ((40, 6), (120, 5), (378, 17)):
        rs = np.random.RandomState(o)
        A0 = 0.95 * np.eye(o) + 0.05 * rs.randn(o, o) / math.sqrt(o)
        B0 = 0.1 * rs.randn(d, o)
        assert imb(A0, B0) <= 10
        assert imb(rs.randn(o, o), rs.randn(d, o)) <= 5
        A = A0.copy()
        A[10:20, 10:20] *= 2.0 ** 14
        A[10:20, :10] *= 2.0 ** -6
        A[10:20, 20:] *= 2.0 ** -6
        assert imb(A, B0) > 20
        A = A0.copy()
        A[5, :] *= 1e-30
        assert imb(A, B0) <= 10
    assert lib.icem_wide_model_imbalance_log2(0, 1, None, None) == -1


def test_options_are_set_through_the_abi_not_the_environment(lib, monkeypatch):
    """kernel-path selection lives in one table behind icem_set_option; the library reads no
    environment variable (its dynamic symbol table does not even import getenv); tools map ICEM_<NAME> explicitly."""
    import shutil
    import subprocess
    names = L.option_names()
    assert "fuse_max_rw" in names and "noise_ahead" in names and len(names) >= 20
    assert not any("arith" in n for n in names)   # no option selects an arithmetic: that is per handle
    L.reset_options()
    assert L.get_option("fuse_max_rw") == 8.0 and L.get_option("ahead_tail_frac") == 0.4
    L.set_option("fuse_max_rw", 0)
    assert L.get_option("fuse_max_rw") == 0.0
    with pytest.raises(L.IcemError):
        L.set_option("no_such_option", 1)
    with pytest.raises(L.IcemError):
        L.set_option("fuse_max_rw", float("nan"))
    # the environment alone changes nothing ...
    monkeypatch.setenv("ICEM_FUSE_MAX_RW", "2")
    monkeypatch.setenv("ICEM_GK_ROLLOUT", "thread")
    L.reset_options()
    assert L.get_option("fuse_max_rw") == 8.0 and L.get_option("gk_rollout_thread") == 0.0
    # ... until a TOOL maps it
    done = L.apply_env_options()
    assert done == {"fuse_max_rw": 2.0, "gk_rollout_thread": 1.0}
    assert L.get_option("fuse_max_rw") == 2.0 and L.get_option("gk_rollout_thread") == 1.0
    L.reset_options()
    if shutil.which("nm"):
        syms = subprocess.check_output(["nm", "-D", "--undefined-only", L.lib_path()], text=True)
        assert "getenv" not in syms
        faults = os.path.join(os.path.dirname(L.lib_path()), "libicem_hip_faults.so")
        if os.path.exists(faults):   # the fault-injection twin (tests/test_gpu_exchange_faults.py) is the one binary that does
            assert "getenv" in subprocess.check_output(["nm", "-D", "--undefined-only", faults], text=True)
