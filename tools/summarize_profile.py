#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output dir into profiles/: kernel stats (copied) + per-kernel HBM
traffic from the FETCH_SIZE / WRITE_SIZE passes.  usage: summarize_profile.py gpurun_out/r01-c4 profiles/r01_c4"""
import csv, json, os, shutil, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst), exist_ok=True)
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), dst + "_kernel_stats.csv")
bench = open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1]
open(dst + "_bench_under_rocprof.json", "w").write(bench + "\n")


def counter(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == name:
            nm = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = nm.split("(")[0][:60] + f" grid={row['Grid_Size']}"
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return acc


fetch = counter(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
write = counter(os.path.join(src, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
with open(dst + "_hbm_traffic.md", "w") as f:
    f.write("# HBM traffic per launch from rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n\n")
    f.write("Units: counters are KiB. Per MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 reports half the bytes of a wide\n"
            "coalesced read: the `fetch x2` column applies that correction; WRITE_SIZE is uncalibrated (reported raw).\n\n")
    f.write("| kernel (grid threads) | launches | FETCH_SIZE KiB/launch | fetch x2 MB | WRITE_SIZE KiB/launch | write MB |\n|---|---|---|---|---|---|\n")
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][0] + write[k][0])):
        fv, fn = fetch[k]
        wv, wn = write[k]
        if max(fn, wn) < 5 or "icem" not in k:
            continue
        fa = fv / max(fn, 1)
        wa = wv / max(wn, 1)
        f.write(f"| `{k}` | {max(fn, wn)} | {fa:.1f} | {2 * fa * 1024 / 1e6:.2f} | {wa:.1f} | {wa * 1024 / 1e6:.2f} |\n")
# machine-readable: per kernel family, launch-weighted HBM bytes per launch (reads x2-corrected per the guide, writes raw)
fam = defaultdict(lambda: [0.0, 0.0, 0])
for k in set(fetch) | set(write):
    name = k.split("<")[0].replace("icem::", "")
    fv, fn = fetch[k]
    wv, wn = write[k]
    n = max(fn, wn)
    if n < 5:
        continue
    fam[name][0] += 2 * fv * 1024 / max(fn, 1) * n
    fam[name][1] += wv * 1024 / max(wn, 1) * n
    fam[name][2] += n
# ... and its average duration in the kernel trace (calls-weighted over the template instances of the family)
dur = defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(dst + "_kernel_stats.csv")):
    nm = row["Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("icem::", "")
    name = nm.split("<")[0].split("(")[0]
    dur[name][0] += float(row["AverageNs"]) * int(row["Calls"])
    dur[name][1] += int(row["Calls"])
json.dump({name: {"read_bytes_per_launch": v[0] / v[2], "write_bytes_per_launch": v[1] / v[2], "launches": v[2],
                  "trace_avg_duration_ns": (dur[name][0] / dur[name][1]) if dur[name][1] else None}
           for name, v in fam.items() if v[2]}, open(dst + "_hbm_traffic.json", "w"), indent=1)
print(open(dst + "_hbm_traffic.md").read())

# instruction mix: SQ counters of the two extra passes, per kernel family and launch (sums over all SEs / XCDs)
SQ = {}
for sub, stem in (("pmc_sq1", "s1"), ("pmc_sq2", "s2")):
    path = os.path.join(src, sub, stem + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for row in csv.DictReader(open(path)):
        nm = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("icem::", "")
        name = nm.split("<")[0].split("(")[0]
        if "kernel" not in name:
            continue
        d = SQ.setdefault(name, defaultdict(lambda: [0.0, 0]))
        d[row["Counter_Name"]][0] += float(row["Counter_Value"])
        d[row["Counter_Name"]][1] += 1
if SQ:
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F32",
            "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
            "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES"]
    table = {name: {c: (d[c][0] / d[c][1] if d[c][1] else None) for c in cols} for name, d in SQ.items()}
    for name, t in table.items():
        w = t.get("SQ_WAVES") or 0
        t["valu_per_wave"] = (t["SQ_INSTS_VALU"] / w) if w and t.get("SQ_INSTS_VALU") else None
        t["salu_per_wave"] = (t["SQ_INSTS_SALU"] / w) if w and t.get("SQ_INSTS_SALU") else None
        wc = t.get("SQ_WAVE_CYCLES") or 0
        # SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves (MI355X_MICROARCH.md)
        t["frac_wave_cycles_issuing_valu"] = (t["SQ_ACTIVE_INST_VALU"] / wc) if wc and t.get("SQ_ACTIVE_INST_VALU") else None
        t["frac_wave_cycles_issue_stalled"] = (t["SQ_WAIT_INST_ANY"] / wc) if wc and t.get("SQ_WAIT_INST_ANY") else None
        t["frac_wave_cycles_parked"] = (t["SQ_WAIT_ANY"] / wc) if wc and t.get("SQ_WAIT_ANY") else None
    json.dump(table, open(dst + "_instruction_mix.json", "w"), indent=1)
    with open(dst + "_instruction_mix.md", "w") as f:
        f.write("# Instruction mix per launch (rocprofv3 --pmc, SQ block, two passes; sums over the chip)\n\n")
        f.write("| kernel | waves | VALU / wave | SALU / wave | SMEM | LDS | MFMA f32 MOPS | VALU-issuing share of wave cycles | issue-stalled | parked (waitcnt / barrier) | MFMA-busy cycles |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        fmt = lambda v, p=0: "-" if v is None else (f"{v:.{p}f}")  # noqa: E731
        for name, t in sorted(table.items()):
            f.write(f"| `{name}` | {fmt(t['SQ_WAVES'])} | {fmt(t['valu_per_wave'])} | {fmt(t['salu_per_wave'])} | {fmt(t['SQ_INSTS_SMEM'])} | {fmt(t['SQ_INSTS_LDS'])} | "
                    f"{fmt(t['SQ_INSTS_VALU_MFMA_MOPS_F32'])} | {fmt(t['frac_wave_cycles_issuing_valu'], 3)} | {fmt(t['frac_wave_cycles_issue_stalled'], 3)} | "
                    f"{fmt(t['frac_wave_cycles_parked'], 3)} | {fmt(t['SQ_VALU_MFMA_BUSY_CYCLES'])} |\n")
    print(open(dst + "_instruction_mix.md").read())
