#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + HBM traffic counters + instruction-mix counters of the bench command.
# usage: tools/profile_round.sh r02 [c2|c4|c3]     (summarise with tools/summarize_profile.py gpurun_out/r02-c4 profiles/r02_c4)
set -u
TAG=${1:-r02}; WL=${2:-c2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG-$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 50 --warmup 5 --workload $WL --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
# counters in their own passes (no tracing domains besides kernel dispatch)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $CMD > /dev/null 2> $OUT/pmc_write.log
# instruction mix / which pipe is busy (SQ block: 8 slots per pass)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_sq1 -o s1 -- $CMD > /dev/null 2> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -o s2 -- $CMD > /dev/null 2> $OUT/pmc_sq2.log
find $OUT -name "*.csv" | head -20
