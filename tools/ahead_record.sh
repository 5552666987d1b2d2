#!/bin/bash
# GPU box: the opt-in noise-ahead pipeline against the default path -- equivalence test, us per MPC step per population
# (tools/ahead_bench.py), and a kernel trace + timeline at N = 65536.   usage: tools/ahead_record.sh <dir under gpurun_out>
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/${1:-ahead}
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "noise_ahead" 2>&1 | tail -3
timeout 400 python tools/ahead_bench.py 16384 32768 65536 131072 262144 2>&1 | grep "N=" | tee $OUT/final.txt
cd /tmp && export TMPDIR=/tmp
ICEM_AB_ONLY=ahead rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ahead -o t -- python $GRAFT_REPO_ROOT/tools/ahead_bench.py 65536 > $OUT/trace_ahead.log 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $(find $OUT/trace_ahead -name "*kernel_trace.csv") 60 8 | tee $OUT/timeline_ahead.txt
cp $(find $OUT/trace_ahead -name "*kernel_stats.csv") $OUT/kernel_stats_ahead.csv
