#!/usr/bin/env python3
"""One-rank RCCL smoke test: the nccl (= RCCL) backend initialises bound to the device the way dp.init_from_env does for
WORLD_SIZE > 1, and an all-reduce of a flat fp32 buffer of the LoRA-gradient size (9.04 M floats at r=16) runs and times.
(The build loop has one GPU; the multi-rank path itself is covered by the gloo tests and first runs on the driver's node.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
except TypeError:
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
g = torch.randn(9_043_968, device="cuda")
ref = g.clone()
for _ in range(3):
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
assert torch.equal(g, ref)
dist.barrier()
print(f"RCCL ok: backend={dist.get_backend()} world={dist.get_world_size()} all_reduce(36 MB) {dt * 1e6:.0f} us/call")
dist.destroy_process_group()
