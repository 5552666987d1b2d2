#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04-split-pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/dbg/wide_sweep.py 16384"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -o s -- $CMD > /dev/null 2> $OUT/p1.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p2 -o s -- $CMD > /dev/null 2> $OUT/p2.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_WAVE32_LDS --output-format csv -d $OUT/p3 -o s -- $CMD > /dev/null 2> $OUT/p3.log
python - <<PY
import csv, glob, collections
for d in ("p1","p2","p3"):
    acc = collections.defaultdict(lambda: [0.0,0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "rollout_wide_split" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k,(v,n) in sorted(acc.items()): print(d, k, "%.4g per launch" % (v/max(n,1)), n)
PY
tail -3 $OUT/p3.log
