"""world_size-1 RCCL probe of the exact collective the sharded bench issues: in-place all_gather_into_tensor whose
input is this rank's slice of the output."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
K, rs = 10, 182
records = torch.arange(K * rs, dtype=torch.float32, device="cuda").view(K, rs).clone()
want = records.clone()
mine = records[0:K]
for _ in range(3):
    dist.all_gather_into_tensor(records, mine)
torch.cuda.synchronize()
assert torch.equal(records, want)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("in-place all_gather_into_tensor over RCCL ok", float(t))
dist.destroy_process_group()
