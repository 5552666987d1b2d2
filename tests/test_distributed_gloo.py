"""N>1 path on CPU: world_size-2/3 gloo runs of the product's sharding + record exchange
(icem_amd.distributed), with the oracle standing in for the kernels.  Every rank must end with
the replicated global elite set of the single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from icem_amd.distributed import exchange_records, shard_range
from oracle import icem_oracle as O


def test_shard_range_partitions_exactly():
    for n in (0, 1, 2, 7, 128, 4096, 3276, 65536, 1676):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert all(hi - lo <= -(-n // w) for lo, hi in spans)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, K, h, d, o, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, cost = O.SyntheticModel.make(o, d), O.CostSpec.halfcheetah(o)
        obs = 0.1 * np.random.RandomState(0).randn(o)
        mean = np.zeros((h, d))
        std = 0.5 * np.ones((h, d))
        low, high = -np.ones(d), np.ones(d)
        hd, rs = h * d, h * d + 2
        for it, n_it in enumerate(O.population_sizes(N, K, 1.25, 3)):
            lo, hi = shard_range(n_it, rank, world)
            zr, zi = O.philox_white_noise(seed, it, hi - lo, d, h, first_index=lo)  # keyed by GLOBAL index
            act = O.sample_action_sequences(mean, std, low, high, 0.25, zr, zi)
            costs = O.rollout_costs(model, cost, obs, act)
            idx = O.topk_sorted(costs, min(K, hi - lo))
            records = torch.zeros((world * K, rs), dtype=torch.float64)
            mine = records[rank * K:(rank + 1) * K]
            mine[:, 0] = float("inf")
            mine[:, 1] = float(np.iinfo(np.int32).max)
            for r, li in enumerate(idx):
                mine[r, 0] = costs[li]
                mine[r, 1] = lo + li
                mine[r, 2:] = torch.from_numpy(act[li].reshape(-1))
            exchange_records(records, K, rank, world)          # the ONE collective of the iteration
            rec = records.numpy()
            order = np.lexsort((rec[:, 1], rec[:, 0]))[:K]     # (cost, gidx) ascending
            elites = rec[order, 2:].reshape(K, h, d)
            mean, std = O.refit(elites, mean, std, 0.1)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), mean=mean, std=std, gidx=rec[order, 1], cost=rec[order, 0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_matches_single_process(tmp_path, world):
    N, K, h, d, o, seed = 257, 10, 12, 4, 17, 5
    mp.spawn(_worker, args=(world, _free_port(), N, K, h, d, o, seed, str(tmp_path)), nprocs=world, join=True)
    # single process reference
    model, cost = O.SyntheticModel.make(o, d), O.CostSpec.halfcheetah(o)
    obs = 0.1 * np.random.RandomState(0).randn(o)
    mean, std = np.zeros((h, d)), 0.5 * np.ones((h, d))
    for it, n_it in enumerate(O.population_sizes(N, K, 1.25, 3)):
        zr, zi = O.philox_white_noise(seed, it, n_it, d, h)
        act = O.sample_action_sequences(mean, std, -np.ones(d), np.ones(d), 0.25, zr, zi)
        costs = O.rollout_costs(model, cost, obs, act)
        idx = O.topk_sorted(costs, K)
        mean, std = O.refit(act[idx], mean, std, 0.1)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["gidx"].astype(np.int64), idx)           # bit-exact global elite indices
        assert np.array_equal(z["cost"], costs[idx])
        assert np.array_equal(z["mean"], mean) and np.array_equal(z["std"], std)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the driver's multi-GPU command is exactly
    that; the reference's ParallelGroundTruthModel forks its own workers, icem/models/gt_par_model.py:26-37): every rank
    gets RANK / WORLD_SIZE / MASTER_*, rendezvous works, rank 0 alone prints ONE line.  (Dry run: no GPU work.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ICEM_BENCH_DRYRUN"] = "1"
    for n in (1, 2, 3):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        j = json.loads(lines[0])
        assert j["n_gpus"] == n and j["rank_sum"] == n * (n + 1) / 2
        assert (j["launched_by"] == "direct") == (n == 1)


# ---------------------------------------------------------------------------------------------
# bench.py's agreement on a failed records path (world 2, gloo, a stand-in planner: no GPU)
# ---------------------------------------------------------------------------------------------

class _StandInPlanner:
    """What bench.timed_steps needs of a planner: plan_step_resident raises IcemError from a given step on while the
    path is "ipc" (a bounded device-side wait that ran out), degrade_exchange steps down."""

    def __init__(self, rank, fail_rank, fail_from):
        from types import SimpleNamespace
        self.cfg = SimpleNamespace(rank=rank, world=2)
        self.path, self.calls, self.fail_rank, self.fail_from = "ipc", 0, fail_rank, fail_from
        self._exchange = False    # (no status word to read: the failure shows as an exception)
        self.leaves_block_safely = True   # the in-library exchange has no collective inside a step (bench.timed_steps)
        self.degraded = []

    def plan_step_resident(self):
        from icem_amd import _lib as L
        self.calls += 1
        if self.path == "ipc" and self.cfg.rank == self.fail_rank and self.calls >= self.fail_from:
            raise L.IcemError(-4, "in-library exchange: a wait for a peer's elite records timed out")

    def degrade_exchange(self):
        self.path = "host"
        self.degraded.append(self.path)
        return self.path


def _agree_worker(rank, world, port, out_dir):
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        # rank 1 alone starts failing in the middle of the first timed block: BOTH ranks must see RecordsPathFailed ...
        pl = _StandInPlanner(rank, fail_rank=1, fail_from=9)
        raised = False
        try:
            bench.timed_steps(pl, 6, 5, world, {})
        except bench.RecordsPathFailed:
            raised = True
        # ... and with the fallback loop both step down once and finish the measurement on the next path
        pl2 = _StandInPlanner(rank, fail_rank=1, fail_from=9)
        spread = {}
        el = bench.timed_with_fallback(pl2, 6, 5, world, spread)
        np.savez(os.path.join(out_dir, f"agree{rank}.npz"), raised=raised, degraded=len(pl2.degraded), path=pl2.path, el=el,
                 noted=spread.get("records_path_degraded_to", []) == ["host"])
    finally:
        dist.destroy_process_group()


def test_bench_ranks_agree_on_a_failed_records_path_and_step_down_together(tmp_path):
    """A rank whose MPC step raises stays inside the collective pattern of bench.py's timed blocks: the all-reduce that closes
    the block carries the failure to every rank, all of them step down (the healthy rank too) and the measurement starts over
    -- the run-time leg of the records' path drills, on CPU (the GPU form: tests/test_gpu_exchange_faults.py)."""
    port = _free_port()
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        z = np.load(tmp_path / f"agree{r}.npz")
        assert bool(z["raised"]) and int(z["degraded"]) == 1 and str(z["path"]) == "host" and bool(z["noted"])
        assert float(z["el"]) > 0


class _DivergingPlanner(_StandInPlanner):
    """Nothing raises, but while the path is "ipc" one rank's refit comes out different (a stale record): the checksum of
    mean | std that rides in the block's all-reduce is what notices."""

    def __init__(self, rank, diverge_rank):
        super().__init__(rank, fail_rank=-1, fail_from=1 << 30)
        self.diverge_rank = diverge_rank
        self.mean = torch.zeros(30, 6)
        self.std = torch.full((30, 6), 0.5)

    def plan_step_resident(self):
        self.calls += 1
        self.mean = torch.full((30, 6), 0.001 * self.calls)
        if self.path == "ipc" and self.cfg.rank == self.diverge_rank:
            self.mean[3, 2] += 1e-7

    def degrade_exchange(self):
        self.calls = 0
        return super().degrade_exchange()


def _checksum_worker(rank, world, port, out_dir):
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        pl = _DivergingPlanner(rank, diverge_rank=1)
        raised = False
        try:
            bench.timed_steps(pl, 6, 5, world, {})
        except bench.RecordsPathFailed:
            raised = True
        pl2 = _DivergingPlanner(rank, diverge_rank=1)
        spread = {}
        el = bench.timed_with_fallback(pl2, 6, 5, world, spread)
        same = _DivergingPlanner(rank, diverge_rank=-1)   # ranks that agree: no failure, the field says so
        spread_ok = {}
        bench.timed_with_fallback(same, 6, 5, world, spread_ok)
        np.savez(os.path.join(out_dir, f"sum{rank}.npz"), raised=raised, path=pl2.path, el=el,
                 noted=spread.get("records_path_degraded_to", []) == ["host"], disagreed=spread.get("blocks_the_ranks_disagreed_after", 0),
                 agree=bool(spread.get("ranks_agree_after_every_block")), clean=bool(spread_ok.get("ranks_agree_after_every_block")) and
                 "records_path_degraded_to" not in spread_ok and same.path == "ipc")
    finally:
        dist.destroy_process_group()


def test_bench_notices_ranks_whose_distributions_differ_and_steps_down(tmp_path):
    """Records that arrive stale or torn on one rank raise nothing: the block's all-reduce also carries a checksum of every
    rank's mean | std (bench.distribution_checksum); ranks that differ end the block like a failed path -- all step down, the
    measurement starts over, the line says how often (`timed_region.blocks_the_ranks_disagreed_after`)."""
    port = _free_port()
    mp.spawn(_checksum_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        z = np.load(tmp_path / f"sum{r}.npz")
        assert bool(z["raised"]) and str(z["path"]) == "host" and bool(z["noted"]) and int(z["disagreed"]) == 1
        assert bool(z["agree"]) and bool(z["clean"]) and float(z["el"]) > 0
