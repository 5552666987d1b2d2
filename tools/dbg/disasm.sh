#!/bin/bash
# Disassemble the gfx950 code of one built translation unit: tools/dbg/disasm.sh k_rollout_ahead  ->  /tmp/isa/<unit>.s (+ per-kernel resource notes in <unit>.notes)
set -e
U=${1:?unit}; LLVM=/opt/rocm/lib/llvm/bin; ROOT=$(cd $(dirname $0)/../.. && pwd)
mkdir -p /tmp/isa; cd /tmp/isa
O=$(ls $ROOT/icem_amd/csrc/_obj/$U.*.o | head -1)
$LLVM/llvm-objcopy --dump-section=.hip_fatbin=$U.fatbin $O
$LLVM/clang-offload-bundler --unbundle --type=o --input=$U.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$U.co
$LLVM/llvm-objdump -d --demangle $U.co > $U.s
$LLVM/llvm-readelf --notes $U.co > $U.notes
echo /tmp/isa/$U.s
