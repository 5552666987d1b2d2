#!/bin/bash
# Per-kernel average durations of tools/quick_bench.py under rocprofv3 (run on the GPU box through gpurun).
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/kt -o t -- python $ROOT/tools/quick_bench.py "$@" > $ROOT/gpurun_out/kt.log 2>&1
grep '^N=' $ROOT/gpurun_out/kt.log
grep -E "icem" $ROOT/gpurun_out/kt/t_kernel_stats.csv | sed -E 's/"void //; s/icem::\(anonymous namespace\):://; s/icem:://' | awk -F'",' '{split($2,a,","); printf "%-60s calls=%s avg_us=%.2f\n", substr($1,2,60), a[1], a[3]/1000}'
