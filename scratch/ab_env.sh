cd /tmp; export TMPDIR=/tmp
for v in BASE "$@"; do
  if [ "$v" = BASE ]; then E=""; else E="$v=1"; fi
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abe_$v -o t -- python $GRAFT_REPO_ROOT/tools/quick_bench.py 65536,1,-1 > $GRAFT_REPO_ROOT/gpurun_out/abe_$v.log 2>&1
  echo "== $v: $(grep '^N=' $GRAFT_REPO_ROOT/gpurun_out/abe_$v.log | tr '\n' ' ')"
  grep -E "rollout_mfma|sample_|merge_single" $GRAFT_REPO_ROOT/gpurun_out/abe_$v/t_kernel_stats.csv | sed -E 's/.*namespace\)::([a-z_]+).*\)",([0-9]+),([0-9]+),([0-9.]+),.*/\1 calls=\2 avg_ns=\4/'
done
