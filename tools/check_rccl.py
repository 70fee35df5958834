"""Sanity check of the N>1 plumbing on one GPU: torchrun-style env, RCCL process group of size 1, the helpers of
parallel.py, and a 2-step bench through torch.distributed.run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from czk_amd import parallel
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("rccl all_reduce ok", t.tolist(), "max_over_ranks", parallel.max_over_ranks(1.5, device="cuda"),
      "gather", parallel.all_gather_shares(torch.arange(4, device="cuda")).shape)
parallel.barrier(torch.cuda.synchronize)
dist.destroy_process_group()
