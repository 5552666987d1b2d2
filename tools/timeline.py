"""Timeline of one MPC step from a rocprofv3 kernel trace (t_kernel_trace.csv): per kernel its queue, start offset and
duration -- which launches overlapped.   usage: python tools/timeline.py <kernel_trace.csv> [skip_steps] [n_kernels]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "icem" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
# start at a merge_single_kernel (end of a step) after `skip` steps
idx = [i for i, r in enumerate(rows) if "merge_single_kernel" in r["Kernel_Name"] or "merge_noise_kernel" in r["Kernel_Name"]]
i0 = idx[min(skip, len(idx) - 3)] + 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + n]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void icem::", "").split("(")[0][:46]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r['Queue_Id']:>2s} {nm:48s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} x {r['Workgroup_Size_X']:>4s}  "
          f"start {s / 1e3:8.1f}  end {e / 1e3:8.1f}  dur {(e - s) / 1e3:6.1f} us")
