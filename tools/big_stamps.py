#!/usr/bin/env python3
"""In-kernel cycle stamps of gemm_big.hip's k-loop (CVFT_BIG_STAMP=1 CVFT_GEMM_BIG=2): one wave of each row, block 8, k-tiles 8..11."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
M, N, K = 5328, 3072, 1024
x, w, o = torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / 32, torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5):
    HF.gemm(x, w, out=o)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
f = HF.lib().cvft_debug_big_stamps
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
t00 = buf[0]
for wr in range(2):
    print(f"row {wr}:  slot: start  reads_issued  lgkm0  barA_out | mfma_issued  vm0  barB_out   (cycles since row0 slot0 start; deltas)")
    for s in range(8):
        v = [buf[wr * 128 + s * 8 + i] for i in range(7)]
        rel = [int(t - t00) for t in v]
        print(f"  kt{8 + s // 2} ks{s % 2}: " + " ".join(f"{r:7d}" for r in rel) + "   d: " + " ".join(f"{rel[i + 1] - rel[i]:5d}" for i in range(6)))
