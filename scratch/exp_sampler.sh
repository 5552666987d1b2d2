cd /tmp; export TMPDIR=/tmp
for v in EMPTY RNGONLY; do
  if [ $v = BASE ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exps_$v -o t -- python $GRAFT_REPO_ROOT/tools/quick_bench.py 65536,1,-1 > /dev/null 2>&1
  echo "$v: $(grep sample_folded $GRAFT_REPO_ROOT/gpurun_out/exps_$v/t_kernel_stats.csv | cut -d, -f2-4)"
done
