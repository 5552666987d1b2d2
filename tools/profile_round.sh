#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + HBM traffic counters of the bench command.
# usage: tools/profile_round.sh r01 [c2|c4]
set -u
TAG=${1:-r01}; WL=${2:-c2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG-$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 50 --warmup 5 --workload $WL --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
# counters in their own passes (no tracing domains besides kernel dispatch)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $CMD > /dev/null 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
