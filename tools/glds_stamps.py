#!/usr/bin/env python3
"""In-kernel cycle stamps of the default 128x128 LDS-DMA GEMM's k-loop (run with CVFT_GLDS_BIG=15): waves 0 and 7 of block
100, k-tiles 4..11.  Columns: loop top, after the vmcnt wait, after the barrier, after the DMA issue, after reads + MFMAs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (5328, 4096, 1024))]
x, w, o = torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / 32, torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5):
    HF.gemm(x, w, out=o)
torch.cuda.synchronize()
print(HF.lib().cvft_gemm_last_kernel().decode())
buf = (ctypes.c_ulonglong * 256)()
assert HF.lib().cvft_debug_glds_stamps(buf) == 0
t00 = buf[0]
for wv, name in ((0, "wave 0"), (1, "wave 7")):
    print(f"{name}: top  vm_done  bar_out  dma_issued  computed   | deltas: vmwait barrier dma compute | period")
    prev = None
    for s in range(8):
        v = [int(buf[wv * 128 + s * 8 + i] - t00) for i in range(5)]
        per = "" if prev is None else f"{v[0] - prev:6d}"
        prev = v[0]
        print(f"  kt{4 + s}: " + " ".join(f"{r:7d}" for r in v) + "   | " + " ".join(f"{v[i + 1] - v[i]:6d}" for i in range(4)) + " | " + per)
