"""bench.py's arithmetic on recorded numbers (no GPU): the roofline fractions are ALGORITHMIC (bytes / flops the path needs,
not the products a split arithmetic executes), event times that exceed the step they were taken from are flagged instead of
scaled down, the loop's own floors come out of the metric's definition (8 d + 8 / h bytes per traj-step)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_fit_to_step_scales_up_only(bench):
    prof = {"sample_rollout": (0.050, 5, 1000), "merge_refit": (0.006, 1, 10)}
    out, fit = bench.fit_to_step(prof, 1, 0.070)          # the kernels sum to 0.056 ms of a 0.070 ms step: gaps charged pro rata
    assert fit["fits"] and abs(sum(v[0] for v in out.values()) - 0.070) < 1e-12
    out, fit = bench.fit_to_step(prof, 1, 0.040)          # ... to MORE than the step: left alone, flagged
    assert not fit["fits"] and out["sample_rollout"][0] == 0.050
    r = bench.roofline_of(out, bench.WORKLOADS["c2"], None, fit, 0)
    assert r["valid"] is False


def test_roofline_is_algorithmic_and_carries_the_floors(bench):
    w = bench.WORKLOADS["c4"]
    units = 220301 * 30 // 5                               # traj-steps of one launch
    prof = {"sample_rollout": (0.030, 1, units)}
    for arith in (0, 1):
        r = bench.roofline_of(prof, w, None, None, arith)
        bytes_per = 8 * 6 + 8 / 30
        assert abs(r["achieved"] - units * bytes_per / 30e-6 / 1e9) < 1e-6 * r["achieved"]
        a = r["attainable"]
        assert abs(a["hbm_floor_us"] - units * bytes_per / 8e12 * 1e6) < 1e-9
        assert a["attainable_us"] == max(a["hbm_floor_us"], a["alu_floor_us"])
        assert abs(a["frac_of_attainable"] - a["attainable_us"] / 30.0) < 1e-12
    # exact f32: every flop on the 157.3 TFLOP/s pipe -- the ALU floor is above the HBM floor; fp16 planes: below it
    assert bench.roofline_of(prof, w, None, None, 0)["binding_roof"] == "f32-alu"
    assert bench.roofline_of(prof, w, None, None, 1)["binding_roof"] == "hbm"
    # the wide GEMM rollout: `frac` = algorithmic flops over the dense fp16 peak, the executed products beside it
    w3 = bench.WORKLOADS["c3"]
    r3 = bench.roofline_of({"rollout_cost": (0.5, 1, 16384 * 30)}, w3, None, None, 0)
    assert r3["bound"] == "mfma" and abs(r3["executed_frac"] - 3 * r3["frac"]) < 1e-12
    assert abs(r3["frac"] - r3["achieved"] / 2500.0) < 1e-12 and r3["vs_exact_f32_matrix_peak"] > r3["frac"]


def test_loop_floor_matches_the_metric_definition(bench):
    w = bench.WORKLOADS["c2"]
    f = bench.loop_floor_ms(w, 13764, 1)
    assert abs(f["hbm_floor_ms"] - 13764 * 30 * (48 + 8 / 30) / 8e12 * 1e3) < 1e-12
    assert f["attainable_ms"] == max(f["hbm_floor_ms"], f["alu_floor_ms"])
