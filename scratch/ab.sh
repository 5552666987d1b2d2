cd /tmp; export TMPDIR=/tmp
for v in BASE $@; do
  if [ $v = BASE ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_$v -o t -- python $GRAFT_REPO_ROOT/tools/quick_bench.py > $GRAFT_REPO_ROOT/gpurun_out/ab_$v.log 2>&1
  echo "== $v: $(grep '^N=' $GRAFT_REPO_ROOT/gpurun_out/ab_$v.log | tr '\n' ' ')"
  grep -E "rollout_mfma|sample_folded|merge_single" $GRAFT_REPO_ROOT/gpurun_out/ab_$v/t_kernel_stats.csv | sed -E 's/.*namespace\)::([a-z_]+).*\)",([0-9]+),([0-9]+),([0-9.]+),.*/\1 calls=\2 avg_ns=\4/'
done
