#!/usr/bin/env python3
"""Graph-timed LoRA-gradient slab kernel (cvft_lora_rank_partial) at the step's shapes, checked against torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.binding import lib, check, dt, ptr, stream
from tools.bench_kernels import timeit

dev = "cuda"
torch.manual_seed(0)
for M, Cn, r, tr in [(4000, 256, 16, 0), (4000, 512, 16, 1), (8000, 256, 16, 0), (4640, 512, 16, 0), (5328, 1024, 16, 0),
                     (5328, 1024, 16, 1), (5328, 4096, 16, 1), (640, 1024, 16, 0), (4000, 256, 32, 0), (1000, 264, 64, 1)]:
    x = torch.randn(M, Cn, device=dev, dtype=torch.bfloat16)
    v = torch.randn(M, r, device=dev, dtype=torch.bfloat16)
    rpb, ns = HF.LoraGradSink.plan(M, Cn)
    if len(sys.argv) > 1:
        rpb = int(sys.argv[1]); ns = -(-M // rpb)
    ws = torch.zeros(ns * r * Cn, device=dev)
    f = lambda: check(lib().cvft_lora_rank_partial(dt(x), M, Cn, r, ptr(x), x.stride(0), ptr(v), v.stride(0), ptr(ws), tr, rpb, stream()), "rp")
    f()
    got = ws.view(ns, -1).sum(0).view((Cn, r) if tr else (r, Cn))
    ref = v.float().t() @ x.float()
    ref = ref.t() if tr else ref
    err = float((got - ref).norm() / ref.norm())
    t = timeit(f)
    print(f"M{M:5d} C{Cn:5d} r{r:3d} tr{tr} rpb{rpb:5d} ns{ns:4d}: {t:7.1f} us  {M * Cn * 2 / t / 1e3:7.1f} GB/s  rel {err:.1e}")
