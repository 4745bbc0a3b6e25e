#!/usr/bin/env python3
"""In-kernel cycle shares of the bf16 attention forward (diagnostic build of the library with -DA32_STAMPS):
   bash tools/build_stamps.sh && CVFT_LIB_PATH=.../libcvft_stamps.so python tools/attn_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF, binding as cb

NAMES = ["prologue", "next QK(+band,skew)", "max", "exp", "rescale+pack", "PV", "bar1/idle", "pf wait+store", "bar2", "epilogue"]
dev, dt = "cuda", torch.bfloat16
lib = cb.lib()
lib.cvft_debug_attn_stamps.argtypes = [ctypes.c_void_p]
for name, B, H, L, rel, causal in [("estimator T250", 16, 8, 250, False, False), ("estimator T500", 16, 8, 500, False, False),
                                   ("flow enc  L290", 16, 8, 290, True, False), ("llm       L333", 16, 16, 333, True, True)]:
    d = H * 64
    q, k, v = (torch.randn(B * L, d, device=dev, dtype=dt) for _ in range(3))
    ln = torch.full((B,), L, device=dev, dtype=torch.int32)
    nblk = (L + 127) // 128 * H * B
    buf = torch.zeros(nblk * 4 * 12, dtype=torch.int64, device=dev)
    assert lib.cvft_debug_attn_stamps(buf.data_ptr()) == 0
    with torch.no_grad():
        for _ in range(3):
            if rel:
                p = torch.randn(2 * L - 1, d, device=dev, dtype=dt)
                bu, bv = torch.randn(H, 64, device=dev) * 0.1, torch.randn(H, 64, device=dev) * 0.1
                HF.attn_relpos(q, k, v, p, bu, bv, B, H, L, ln, causal, 0.125)
            else:
                HF.attn_bias(q, k, v, B, H, L, ln, 0.125)
    torch.cuda.synchronize()
    s = buf.view(nblk * 4, 12).double().cpu()
    tot = s.sum(1)
    act = tot > 0
    print(f"{name}: {int(act.sum())} waves, mean lifetime {tot[act].mean():8.0f} cycles, max {tot.max():8.0f}")
    for i, n in enumerate(NAMES):
        print(f"    {n:16s} mean {s[act, i].mean():8.0f}  ({100 * s[act, i].sum() / tot[act].sum():5.1f} %)   max-wave {s[:, i].max():8.0f}")
    lib.cvft_debug_attn_stamps(None)
