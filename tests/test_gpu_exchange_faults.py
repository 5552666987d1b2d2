"""First contact with more than one GPU, drilled on whatever this box has: `python bench.py --gpus 2` under
injected failures of the in-library elite exchange (ICEM_XCHG_FAIL, csrc/exchange.hip) must END -- on the next path of
exchange.hip -> collective.hip (icem_allgather_elites) -> host-driven all-gather -- and say in its one JSON line which path
carried the records (`exchange.ran`) and why the first one did not (`exchange.in_library_exchange_error`):

  connect   one rank cannot map its peers' blocks (hipIpcOpenMemHandle across devices): all ranks step down at set-up
  selftest  the blocks map but a rank's payload check fails (a coarse-grained block shadowed by the owner's L2)
  timeout   the exchange works, then a rank stops publishing mid-run: every rank's bounded wait runs out on the device, the
            next icem_plan_step_sharded raises, the ranks agree through the collective that closes every timed block and
            step down together at RUN time; the measurement starts over

On a one-GPU box both ranks share the device (gloo rendezvous, disjoint CU slices) and RCCL refuses two ranks on one GPU:
the run ends on "host"; with a GPU per rank it ends on "rccl".  Reference analogue of the path: the pipe gather of
icem/models/gt_par_model.py:77-94."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(fault):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ICEM_XCHG_FAIL")}
    if fault:
        # the product library carries no fault injection: the drills load its twin built with -DICEM_FAULT_INJECTION
        # (icem_amd/build.py: exchange.hip compiled once more, everything else the same objects)
        env["ICEM_HIP_LIB"] = os.path.join(ROOT, "icem_amd", "libicem_hip_faults.so")
        env["ICEM_XCHG_FAIL"] = fault
        env["ICEM_XCHG_MAX_POLLS"] = "20000"   # a lost peer costs milliseconds, not the default's seconds (bench.py maps it onto icem_set_option)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
                        "--no-cpu-baseline", "--no-also"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_without_a_fault_the_in_library_exchange_carries_the_records():
    j, _ = _bench(None)
    ex = j["exchange"]
    assert ex["ran"] == "ipc" and ex["timeouts"] == 0 and ex["host_collectives_in_timed_loop"] == 0
    assert "records_path_degraded_to" not in j["timed_region"]


@pytest.mark.parametrize("fault", ["connect", "selftest:0", "timeout:1:40"])
def test_every_injected_failure_ends_on_a_fallback_path_and_says_so(fault):
    j, err = _bench(fault)
    ex = j["exchange"]
    want = "rccl" if torch.cuda.device_count() >= 2 else "host"
    assert ex["ran"] == want, (ex, err[-1500:])
    assert ex["in_library_exchange_error"], ex
    assert j["value"] > 0 and j["n_gpus"] == 2
    if fault.startswith("timeout"):
        # the exchange was connected and passed its self-test: the failure came at run time and every rank stepped down
        assert j["timed_region"]["records_path_degraded_to"] == [want], j["timed_region"]
        assert "timed out" in ex["in_library_exchange_error"]
    else:
        assert "records_path_degraded_to" not in j["timed_region"]


@pytest.mark.parametrize("world,scale", [(4, None), (8, None), (8, "32.768")])
def test_four_and_eight_processes_exchange_their_records_through_the_library(world, scale):
    """The in-library exchange between `world` real processes (one IPC handle per peer, world - 1 peer blocks mapped, flags of
    every peer polled) -- on this box all of them on the same GPU, so at a population whose workgroups are resident together
    (tools/dbg/shared_gpu_worlds.py): HalfCheetah and Door shapes, every rank bit for bit the single-process run, nobody
    timed out.  Two ranks are what the other tests of this file and of test_gpu_parity.py run.
    scale 32.768: BASELINE.json configs[3] as written -- N = 65 536 global over 8 ranks (and Door at 49 152) -- with every
    rank on its own eighth of the CUs (HSA_CU_MASK), so that the ranks' launches are resident side by side as on a node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "ICEM_XCHG_FAIL", "ICEM_XCHG_MAX_POLLS",
                                                            "HSA_CU_MASK", "ICEM_SHARED_SCALE", "ICEM_SHARED_SLICES")}
    if scale:
        env.update(ICEM_SHARED_SCALE=scale, ICEM_SHARED_SLICES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "shared_gpu_worlds.py"), str(world)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(f"world {world} ")]
    assert len(lines) == 2 and all("identical to" in ln for ln in lines), r.stdout[-2000:]
    assert all(ln.count("(1, 0)") == world for ln in lines), lines   # connected, status word 0 on every rank
    assert "all identical" in r.stdout
