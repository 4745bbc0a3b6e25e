#!/usr/bin/env python3
"""cvft_ce_fwd / cvft_ce_bwd at the LLM's shapes (rows x 4 097 classes in a 4 160-column buffer) inside a hipGraph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.functional import ptr, stream, check, lib, dt

dev, bf = "cuda", torch.bfloat16
REPS = 16


def timeit(call):
    call()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


for n in (2664, 5328):
    V, pitch = 4097, 4160
    base = torch.zeros(n, pitch, device=dev, dtype=bf)
    base[:, :V] = torch.randn(n, V, device=dev).to(bf) * 2
    lg = base[:, :V]
    tg = torch.randint(0, V, (n,), device=dev, dtype=torch.int32)
    out3 = torch.zeros(3, device=dev)
    lse = torch.empty(n, device=dev)
    gs = torch.ones(1, device=dev)
    dl = torch.zeros(n, pitch, device=dev, dtype=bf)
    tf = timeit(lambda: check(lib().cvft_ce_fwd(dt(lg), n, V, ptr(lg), pitch, ptr(tg), ptr(out3), ptr(lse), 0.0, stream()), "f"))
    tb = timeit(lambda: check(lib().cvft_ce_bwd(dt(lg), n, V, ptr(lg), pitch, ptr(tg), ptr(lse), ptr(gs), ptr(dl), pitch, 0.0, stream()), "b"))
    mb = n * V * 2 / 1e6
    print(f"rows {n}: ce_fwd {tf:6.1f} us ({mb / tf * 1e3 / 1e3:5.2f} TB/s)   ce_bwd {tb:6.1f} us ({2 * mb / tb * 1e3 / 1e3:5.2f} TB/s)")
