#!/bin/bash
# GPU box: kernel trace of the default (sampler + rollout) path at N = 65536 and N = 4096 through tools/ahead_bench.py / quick_bench
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/${1:-pair}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ICEM_AB_ONLY=pair rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pair -o t -- python $GRAFT_REPO_ROOT/tools/ahead_bench.py 65536 > $OUT/trace_pair.log 2>&1
grep "N=" $OUT/trace_pair.log
grep -E "icem" $(find $OUT/trace_pair -name "*kernel_stats.csv") | sed -E 's/"void //; s/icem::\(anonymous namespace\):://; s/icem:://' | awk -F'",' '{split($2,a,","); printf "%-70s calls=%s avg_us=%.2f\n", substr($1,2,70), a[1], a[3]/1000}'
