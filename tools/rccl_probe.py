#!/usr/bin/env python3
"""Sanity probe for the RCCL leg of bench.py on a 1-GPU box: a one-rank "nccl" process group, the barrier and the
max-over-ranks all-reduce that bench.py uses for N > 1 (RCCL refuses two ranks on one device, so the multi-rank case is
rehearsed with gloo instead: UVAD_DIST_BACKEND=gloo torchrun --nproc-per-node 2 bench.py --gpus 2)."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvad_amd import dist as udist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
udist.barrier()
print("max_over_ranks:", udist.max_over_ranks(1.25, device=dev))
x = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(x)
print("all_reduce ok:", x.tolist(), "backend", dist.get_backend())
dist.destroy_process_group()
