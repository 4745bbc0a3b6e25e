#!/usr/bin/env python3
"""Per-shape table of the tap-GEMM launches of one bench step (HIP events, eager): which shapes carry the time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch

dev = torch.device("cuda", 0)
jm = bench.build("joint", torch.bfloat16, dev, 16, 32)
batch = jm.prepare_batch(synth_batch([500] * 16, seed=1234), dev)


def fwd_bwd():
    out = jm(batch, dev)
    with HF.LoraGradSink():
        out['loss'].backward()


for _ in range(2):
    fwd_bwd()
torch.cuda.synchronize()
HF.PROFILE = []
fwd_bwd()
torch.cuda.synchronize()
tab = {}
for r in HF.PROFILE:
    k = (r["kernel"].replace("gemm_", "").replace("_kernel", ""),) + tuple(r["shape"])
    g = tab.setdefault(k, [0.0, 0.0, 0])
    g[0] += r["start"].elapsed_time(r["end"]) * 1e3
    g[1] += r["flop"]
    g[2] += 1
tot = sum(v[0] for v in tab.values())
print(f"total GEMM time {tot / 1e3:.2f} ms over {sum(v[2] for v in tab.values())} launches")
print(f"{'cfg':28s} {'M':>6s} {'N':>5s} {'K':>5s} taps R   n   us/launch  total_ms  TF/s")
for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{k[0]:28s} {k[1]:6d} {k[2]:5d} {k[3]:5d} {k[4]:3d} {k[5]:3d} {v[2]:4d} {v[0] / v[2]:9.1f} {v[0] / 1e3:8.2f} {v[1] / v[0] / 1e6:7.1f}")
