#!/bin/bash
# EXPERIMENTS R4.11: A/B of chunk placements of the split learned-dynamics launch ON ONE BOX (boxes differ by up to 17 %
# at N = 4096).  Variant libraries are linked from the in-tree objects with icem_rssm_split.hip recompiled under
# -DICEM_RSSM_PLAN=k (ChunkPlan) / -DICEM_RSSM_ROT=1, or from `git show <rev>:icem_amd/csrc/icem_rssm_split.hip`:
#   OBJS=$(ls icem_amd/csrc/_obj/*.o | grep -v icem_rssm_split)
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DICEM_RSSM_PLAN=$k -Iinclude -c icem_amd/csrc/icem_rssm_split.hip -o /tmp/split_$k.o
#   hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/split_$k.o -ldl -o tools/experiments/_plans/libicem_plan$k.so
# usage (on the GPU box): bash tools/experiments/rssm_plans.sh 0 B 0 B     (0 = the in-tree library)
for k in "$@"; do
  if [ $k = 0 ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$PWD/tools/experiments/_plans/libicem_plan$k.so; fi
  echo "== plan $k"
  timeout 120 python tools/dbg/rssm_sweep.py 1024 2048 4096 16384 65536 2>&1 | grep "n="
  timeout 120 python tools/dbg/rssm_stamps.py 1024 2>&1 | grep -E "recurrence done|reward: done|phases"
done
