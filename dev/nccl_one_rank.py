"""RCCL API check on one GPU: a one-rank NCCL group through the same helpers the multi-GPU driver uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from wae_amd.nlevp.distributed import _allgather_dev, allreduce_sum_, gather_rows, rank_world
print("rank/world", rank_world(), "backend", dist.get_backend())
t = torch.arange(1000, dtype=torch.float64, device="cuda:0")
g = _allgather_dev(t, 1)
assert torch.equal(g, t)
b = t.clone(); allreduce_sum_(b); torch.cuda.synchronize(); assert torch.equal(b, t)
tab = gather_rows({0: [1 + 2j, 3 - 1j]}, 1, 2); assert tab[0, 1] == 3 - 1j
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl one-rank ok")
