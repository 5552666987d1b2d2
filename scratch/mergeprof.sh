cd /tmp && export TMPDIR=/tmp
for s in 0 1 2; do
ICEM_MERGE_STOP=$s rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/profm$s -o q -- python $GRAFT_REPO_ROOT/tools/quick_bench.py 4096,5,-1 > /dev/null 2>&1
done
