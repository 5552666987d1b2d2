for k in 0 B 0 B; do
  if [ $k = 0 ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$PWD/tools/experiments/_plans/libicem_plan$k.so; fi
  echo "== $k"
  timeout 120 python tools/dbg/step_time.py 16384 32768 65536 131072 2>&1 | grep "N="
  timeout 120 python tools/sharded_rank_bench.py 8 65536 2>&1 | grep "world=" | cut -c1-60
done
