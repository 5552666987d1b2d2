#!/bin/bash
# One rocprofv3 counter pass over tools/quick_bench.py (run on the GPU box): tools/pmc_pass.sh "<counters>" <quick_bench args...>
# Prints per-kernel averages of every counter.
CTRS="$1"; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/pmc; mkdir -p $ROOT/gpurun_out/pmc
rocprofv3 --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc -o p -- python $ROOT/tools/quick_bench.py "$@" > $ROOT/gpurun_out/pmc/run.log 2>&1
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open("$ROOT/gpurun_out/pmc/p_counter_collection.csv")):
    if "icem" not in row["Kernel_Name"]: continue
    nm = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void icem::", "").split("(")[0][:44]
    k = (nm, row["Grid_Size"], row["Counter_Name"])
    acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (nm, grid, c), (v, n) in sorted(acc.items()):
    print(f"{nm:46s} grid={grid:>8s} {c:32s} {v / n:16.1f}  (n={n})")
PY
