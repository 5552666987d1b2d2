#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by several per cent, so only pairs measured in ONE gpurun
# call mean anything).  A variant library = the in-tree objects with ONE unit recompiled from another revision or with
# other flags, e.g. the round's baseline of k_rollout_ahead.hip (EXPERIMENTS R4.12):
#   mkdir -p /tmp/v/a/b/csrc /tmp/v/a/include && cp icem_amd/csrc/*.h /tmp/v/a/b/csrc/ && cp include/icem_hip.h /tmp/v/a/include/
#   git show <rev>:icem_amd/csrc/k_rollout_ahead.hip > /tmp/v/a/b/csrc/k_rollout_ahead.hip      (+ the headers of <rev> it needs)
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c /tmp/v/a/b/csrc/k_rollout_ahead.hip -o /tmp/v/unit.o
#   hipcc --offload-arch=gfx950 -shared -fPIC $(ls icem_amd/csrc/_obj/*.o | grep -v k_rollout_ahead) /tmp/v/unit.o -ldl -o tools/experiments/_libs/libicem_B.so
# (tools/experiments/_libs/*.so travels to the GPU box like every built .so and stays out of git.)
# usage (on the GPU box): bash tools/experiments/ab_libs.sh "<command>" 0 B 0 B      (0 = the in-tree library)
CMD=$1; shift
for k in "$@"; do
  if [ $k = 0 ]; then unset ICEM_HIP_LIB; else export ICEM_HIP_LIB=$PWD/tools/experiments/_libs/libicem_$k.so; fi
  echo "== $k"
  eval "$CMD" 2>&1 | grep -v amdgpu.ids
done
