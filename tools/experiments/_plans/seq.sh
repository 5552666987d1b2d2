for k in 0 0 0 0 0 0; do
  if [ $k = 0 ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$PWD/tools/experiments/_plans/libicem_plan$k.so; fi
  echo "== $k"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "two_processes or soak_two or emulated" 2>&1 | tail -3
done
