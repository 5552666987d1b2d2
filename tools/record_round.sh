#!/bin/bash
# GPU box (via gpurun): everything the committed profiles/<tag>_* files of a round come from, in one call --
#   the GPU test suite, the rocprofv3 passes of c2 / c4 / c3 (tools/profile_round.sh), the default bench line, the c5
#   bench line, the tool lines and the soaks.  Summarise afterwards, here in the container:
#     for w in c2 c4 c3; do python tools/summarize_profile.py gpurun_out/<tag>-$w profiles/<tag>_$w; done
#     cp gpurun_out/<tag>-rec/*.json gpurun_out/<tag>-rec/*.txt -> profiles/<tag>_*   (tools/record_round.sh prints the list)
#     python tools/doc_numbers.py <tag>
# usage: tools/record_round.sh r03 [quick]
TAG=${1:-r03}; QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG-rec
mkdir -p $OUT
cd $ROOT
rm -f $ROOT/gpurun_out/literal_bounds.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -1 $OUT/pytest_gpu.txt
cp $ROOT/gpurun_out/literal_bounds.jsonl $OUT/literal_bounds.jsonl 2> /dev/null
for w in c2 c4 c3 c5; do timeout 900 bash tools/profile_round.sh $TAG $w > /dev/null 2>&1; done
cd $ROOT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json
timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 > $OUT/c5_bench.json 2> /dev/null
# cost_along_trajectory = best / final on the headline workload, and the reference's other shipped settings (cost-term envs)
for m in best final; do timeout 300 python bench.py --cost-mode $m --no-also --no-cpu-baseline > $OUT/bench_c2_$m.json 2> /dev/null; done
for w in door relocate fpp; do timeout 300 python bench.py --workload $w --no-also --no-cpu-baseline > $OUT/bench_$w.json 2> /dev/null; done
# first-contact drills of the records' path: two ranks on this box's GPU(s), every injected failure must end on a fallback path
{
  for f in none connect selftest:0 timeout:1:40; do
    # (the product library carries no fault injection: the drills load its twin, libicem_hip_faults.so)
    if [ $f = none ]; then unset ICEM_XCHG_FAIL ICEM_HIP_LIB; else export ICEM_XCHG_FAIL=$f ICEM_HIP_LIB=$ROOT/icem_amd/libicem_hip_faults.so; fi
    echo "# ICEM_XCHG_FAIL=$f python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --no-also  ->  exchange / timed_region of the line"
    ICEM_XCHG_MAX_POLLS=20000 timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --no-also 2> /dev/null | grep '^{' | python -c "import sys, json; j = json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': j['ms_per_step'], 'exchange': j['exchange'], 'timed_region': j['timed_region']}))"
  done
  unset ICEM_XCHG_FAIL ICEM_HIP_LIB
  echo "# tools/dbg/shared_gpu_worlds.py 2 4 8: the exchange between 2 / 4 / 8 PROCESSES sharing this GPU, at a population whose workgroups are resident together"
  timeout 300 python tools/dbg/shared_gpu_worlds.py 2 4 8 2>&1 | grep "^world\|^shared"
  echo "# ... every rank on its own slice of the CUs (ICEM_SHARED_SLICES=1), 8 processes: N = 65 536 global (BASELINE configs[3]; Door 49 152), then 65 536 PER RANK (N = 524 288 global; Door 393 216)"
  ICEM_SHARED_SLICES=1 ICEM_SHARED_SCALE=32.768 timeout 300 python tools/dbg/shared_gpu_worlds.py 8 2>&1 | grep "^world\|^shared"
  ICEM_SHARED_SLICES=1 ICEM_SHARED_SCALE=262.144 timeout 600 python tools/dbg/shared_gpu_worlds.py 8 2>&1 | grep "^world\|^shared"
  echo "# python -m torch.distributed.run --nproc-per-node {4,8} bench.py at the bench's population on ONE GPU: the ranks' launches do not fit on the chip together, a rank's bounded wait keeps a peer's pack workgroup off it -> the waits run out, all ranks step down together (a GPU per rank has no such tenant)"
  echo "# tools/dbg/shared_gpu_rank_time.py 2 4 8: real peers against absent ones, one tile per CU on every rank's slice; control = the same processes unsharded side by side"
  timeout 600 python tools/dbg/shared_gpu_rank_time.py 2 4 8 2>&1 | grep "^world"
  echo "# python bench.py --gpus {4,8} (self-launched ranks: each rank of a shared GPU gets its own slice of the CUs, HSA_CU_MASK): the same population, the in-library exchange carries the records"
  for n in 4 8; do
    timeout 600 python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-also 2> /dev/null | grep '^{' | python -c "import sys, json; j = json.loads(sys.stdin.read()); print(json.dumps({'n_gpus': j['n_gpus'], 'ranks_share_a_gpu': j['ranks_share_a_gpu'], 'launched_by': j['launched_by'], 'ms_per_step': j['ms_per_step'], 'exchange': j['exchange'], 'timed_region': j['timed_region']}))"
  done
  echo "# ... and under torch.distributed.run (no CU slices):"
  for n in 4 8; do
    timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-also 2> /dev/null | grep '^{' | python -c "import sys, json; j = json.loads(sys.stdin.read()); print(json.dumps({'n_gpus': j['n_gpus'], 'ranks_share_a_gpu': j['ranks_share_a_gpu'], 'ms_per_step': j['ms_per_step'], 'exchange': j['exchange'], 'timed_region': j['timed_region']}))"
  done
} > $OUT/exchange_fault_drills.txt 2>&1
{
  echo "# tools/controller_latency.py (MpcICemHip.get_action, host observation in, host action out)"
  timeout 120 python tools/controller_latency.py 2>&1 | grep "N="
  echo "# tools/sharded_rank_bench.py 8 <rows per GPU> (one rank of 8, peers absent: ICEM_XCHG_LOOPBACK -- everything but the wire)"
  timeout 120 python tools/sharded_rank_bench.py 8 4096 2>&1 | grep "world="
  timeout 120 python tools/sharded_rank_bench.py 8 65536 2>&1 | grep "world="
  echo "# tools/dbg/step_time.py (us per MPC step, icem_plan_step, resident inputs)"
  timeout 120 python tools/dbg/step_time.py 2048 4096 8192 16384 32768 65536 2>&1 | grep "N="
  echo "# tools/dbg/rssm_sweep.py (icem_rssm_rollout_cost alone; TFLOP/s on the nominal 12-transition count)"
  timeout 120 python tools/dbg/rssm_sweep.py 512 1024 2048 4096 16384 65536 2>&1 | grep "n="
  echo "# tools/dbg/rssm_stamps.py 1024 (tile 0 of the split launch)"
  timeout 120 python tools/dbg/rssm_stamps.py 1024 2>&1 | grep -v amdgpu.ids
  echo "# tools/dbg/stamps.py 4096 5 (the headline's last launch, workgroup 0; ICEM_TILE_ARITH=0: the exact tile's VALU twin)"
  timeout 120 python tools/dbg/stamps.py 4096 5 2>&1 | grep "us from"
  ICEM_TILE_ARITH=0 timeout 120 python tools/dbg/stamps.py 4096 5 2>&1 | grep "us from" | sed 's/^/[exact tile] /'
  echo "# ICEM_TILE_ARITH=0 tools/dbg/step_time.py (the exact f32 tile at the same populations)"
  ICEM_TILE_ARITH=0 timeout 120 python tools/dbg/step_time.py 2048 4096 8192 16384 2>&1 | grep "N="
  echo "# tools/dbg/f64_profile.py / f64_stamps.py (the strict-parity path; ICEM_GK_ROLLOUT=thread ICEM_GK_SELECT=0: its round-4 form)"
  timeout 120 python tools/dbg/f64_profile.py 2>&1 | grep "f64 N"
  ICEM_GK_ROLLOUT=thread ICEM_GK_SELECT=0 timeout 120 python tools/dbg/f64_profile.py 4096 10 2>&1 | grep "f64 N" | sed 's/^/[round-4 form] /'
  timeout 120 python tools/dbg/f64_stamps.py 2>&1 | grep "us from"
  echo "# tools/dbg/hn_terms_time.py (TileHN launches at N = 4096, HIP events incl. the ~4 us bracket: wave arrangements, with / without the term list)"
  timeout 300 python tools/dbg/hn_terms_time.py 2>&1 | grep "terms="
  echo "# tools/dbg/hn_check.py (MPC step of the shipped shapes)"
  timeout 300 python tools/dbg/hn_check.py 2>&1 | grep "MPC step"
  echo "# tools/dbg/ahead_stamps.py 65536 (c4: every role of every launch of an MPC step)"
  timeout 200 python tools/dbg/ahead_stamps.py 65536 2>&1 | grep -v amdgpu.ids
  echo "# round 6: tools/dbg/sharded_stamps.py 8 4096 (one sharded launch itemised, records ranked by the whole workgroup)"
  timeout 120 python tools/dbg/sharded_stamps.py 8 4096 2>&1 | grep -v amdgpu.ids
  echo "# round 6: tools/dbg/xcd_stamps.py / xcd_step_check.py (the one-launch step inside one XCD, option step_xcd = 1: stamps, bit-equality, time)"
  timeout 120 python tools/dbg/xcd_stamps.py 4096 5 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/dbg/xcd_step_check.py 2>&1 | grep -v amdgpu.ids | tail -6
  echo "# round 6: tools/ubench/xcd_barrier (a 32-arrival barrier inside one XCD)"
  (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip 2> /dev/null; timeout 120 ./xcd_barrier)
  echo "# round 6: tools/dbg/batch_host_time.py (icem_plan_step_batch: host enqueue time against the step's GPU time)"
  timeout 300 python tools/dbg/batch_host_time.py 2>&1 | grep "B="
  echo "# round 6: icem_plan_step_batch by launch family (option batch_ahead: noise-ahead launches from 49152 rows on / never)"
  timeout 600 python - <<'PYEOF' 2>&1 | grep "batch_ahead"
import bench
from icem_amd import _lib as L
for ah in (1, 0):
    L.reset_options(); L.set_option("batch_ahead", ah); L.set_option("batch_ahead_min_rows", 0)
    j = bench.measure_batched("c2", batches=(6, 8, 12, 16, 32), solo_ms=0.0611)
    for b, r in j["by_B"].items():
        print("batch_ahead (forced)" if ah else "batch_ahead off", "B", b, round(r["ms_per_batched_mpc_step"] * 1e3, 1), "us per batched step,", round(r["aggregate_vs_solo_steps"], 2), "x the solo steps; uploads", r["argument_uploads_in_timed_steps"])
PYEOF
  echo "# tools/ubench/lds_ring (the learned-dynamics weight ring: registers vs LDS)"
  (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o lds_ring lds_ring.hip 2> /dev/null; timeout 120 ./lds_ring)
} > $OUT/tool_lines.txt 2>&1
if [ -z "$QUICK" ]; then
{
  echo "# tools/dbg/soak_equiv.py 1000 1 (icem_plan_step vs split API, all population sizes)"
  timeout 900 python tools/dbg/soak_equiv.py 1000 1 2>&1 | tail -1
  echo "# tools/dbg/soak_equiv.py 300 2 large (noise-ahead launches vs the sampler + rollout pair)"
  timeout 900 python tools/dbg/soak_equiv.py 300 2 large 2>&1 | tail -1
  echo "# tools/dbg/soak_shards.py 600"
  timeout 900 python tools/dbg/soak_shards.py 600 2>&1 | tail -1
  echo "# tools/dbg/soak_oracle.py 100 3 (device vs float64 oracle, elite sets)"
  timeout 600 python tools/dbg/soak_oracle.py 100 3 2>&1 | tail -1
} > $OUT/soak.txt 2>&1
fi
ls $OUT
