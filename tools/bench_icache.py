#!/usr/bin/env python3
"""Does a kernel run slower when its CODE is cold (other kernels + lots of data traffic between two launches)?
Run under rocprofv3 --kernel-trace; mode 0 = same kernel back to back, 1 = interleaved with other kernels,
2 = interleaved + a 1 GB copy between (flushes L2 / MALL)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
mode = int(sys.argv[1])
dev, dt = "cuda", torch.bfloat16
xs = torch.randn(16, 1024, device=dev, dtype=dt); ws = torch.randn(256, 1024, device=dev, dtype=dt); os_ = torch.empty(16, 256, device=dev, dtype=dt)
xm = torch.randn(4000, 1024, device=dev, dtype=dt); om = torch.empty(4000, 256, device=dev, dtype=dt)
xb = torch.randn(5328, 1024, device=dev, dtype=dt); wb = torch.randn(1024, 1024, device=dev, dtype=dt); ob = torch.empty(5328, 1024, device=dev, dtype=dt)
g, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
big = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.float32) if mode == 2 else None
big2 = torch.empty_like(big) if mode == 2 else None
q = torch.randn(16 * 250, 512, device=dev, dtype=dt)
ln = torch.full((16,), 250, device=dev, dtype=torch.int32)


def body():
    for _ in range(10):
        HF.gemm(xs, ws, out=os_)
        if mode >= 1:
            HF.layernorm(xm, g, b)
            HF.attn_bias(q, q, q, 16, 8, 250, ln, 0.125)
            HF.gemm(xb, wb, out=ob)
        if mode == 2:
            big2.copy_(big)
        HF.gemm(xm, ws, out=om)
        if mode >= 1:
            HF.layernorm(xm, g, b)
            HF.gemm(xb, wb, out=ob)
        if mode == 2:
            big2.copy_(big)


with torch.no_grad():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
