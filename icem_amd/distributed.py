"""N-trajectory sharding across GPUs: pure index arithmetic + ONE all-gather per CEM iteration.

The reference's only data parallelism is a process fan-out of trajectories over
``multiprocessing.Pipe`` (icem/models/gt_par_model.py:77-94: ``np.array_split`` -> send ->
recv).  Here rank r of G owns a contiguous slice of the global trajectory indices, the RNG is
keyed by the *global* index (results do not depend on G), and the only exchange is an
all-gather (RCCL over xGMI when the tensors are on GPUs) of each rank's K sorted candidate
records ``{cost, gidx, actions[h*d]}`` before the replicated refit.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Global trajectory indices ``[lo, hi)`` owned by ``rank`` (must match ``shard_chunk`` in
    csrc/host_common.h): chunks of ceil(n/world), trailing ranks may be short or empty."""
    chunk = -(-n_global // world)
    lo = min(n_global, rank * chunk)
    hi = min(n_global, lo + chunk)
    return lo, hi


def exchange_records(records: torch.Tensor, K: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather in place: ``records`` is ``[world*K, rs]`` and this rank has filled rows
    ``[rank*K, (rank+1)*K)``.  One collective of K*(2+h*d) elements per rank per iteration."""
    if world == 1:
        return records
    mine = records[rank * K:(rank + 1) * K]
    backend = dist.get_backend(group)
    if records.is_cuda and backend != "nccl":
        # no RCCL (e.g. several ranks sharing one GPU in a test): stage the K records through the host
        host = torch.empty(records.shape, dtype=records.dtype)
        dist.all_gather_into_tensor(host, mine.cpu(), group=group)
        records.copy_(host)
        return records
    if not records.is_cuda:
        mine = mine.clone()  # gloo: keep input and output disjoint
    dist.all_gather_into_tensor(records, mine, group=group)
    return records
