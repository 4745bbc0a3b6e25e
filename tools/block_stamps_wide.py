#!/usr/bin/env python3
"""Phase stamps of the WIDE block-tail forward kernel (csrc/block_lean.hip, RT = 2; diagnostic build, never the product library):
    bash tools/build_block_stamps.sh block_lean && CVFT_LIB_PATH=.../libcvft_bfstamps.so python tools/block_stamps_wide.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd import modules as Mo
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb

dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
HF.BLOCK_LEAN = os.environ.get("LEAN", "2")
blk = Mo.BasicTransformerBlock(256, 8, 64, 0.0, "gelu").to(dev)
for p in blk.parameters():
    p.requires_grad_(False)
M = int(os.environ.get("M", 4000))
o = torch.randn(M, 512, device=dev, dtype=dt)
x0 = torch.randn(M, 256, device=dev, dtype=dt)
dy = torch.randn(M, 256, device=dev, dtype=dt)
for _ in range(50):
    if os.environ.get("BWD"):
        oo, xx = o.detach().requires_grad_(True), x0.detach().requires_grad_(True)
        blk._tail(oo, xx, "gelu_erf").backward(dy)
    else:
        with torch.no_grad():
            blk._tail(o, x0, "gelu_erf")
torch.cuda.synchronize()
lib = C.CDLL(cb.LIB_PATH)
buf = (C.c_ulonglong * 32)()
assert lib.cvft_debug_block_stamps(buf) == 0
t = list(buf)
if os.environ.get("BWD"):                               # stamps 16.. of block_tail_wide_bwd_kernel
    names = ["dy / x1 tiles in LDS", "H1(0)", "round 0", "rounds 1 .. nr-2", "last round + H2(nr-1)", "LN backward sums", "exchange", "dx1 -> tile + barrier",
             "dx1 store + Wo^T product", "barrier + do tile + store"]
    idx = [16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27]
    for n, a, b in zip(names, idx[:-1], idx[1:]):
        print(f"{n:28s} +{t[b] - t[a]:7d}")
    print("total", t[27] - t[16])
    sys.exit(0)
base = t[0]
for i, name in ((1, "o, x0 tiles + params in LDS"), (2, "out-proj MFMAs"), (3, "x1 into the x tile"), (4, "LN, x1 store, y tiles"), (5, "G1(0)")):
    print(f"{name:28s} +{t[i] - t[i - 1]:7d}  (at {t[i] - base})")
print(f"round 0 (act beside G1(1))   +{t[8] - t[5]:7d}")
if HF.BLOCK_LEAN == "4":                                # eight waves: rounds of 256 hidden units, nr = 4
    for r in (1, 2):
        a, b, c = (t[8] if r == 1 else t[10]), t[9 + 2 * (r - 1)], t[10 + 2 * (r - 1)]
        print(f"  round {r}: G1(r+1) + half act {b - a:6d}   G2(r-1) + half act + barrier + keep {c - b:6d}")
    print(f"last round + G2(nr-1)        +{t[7] - t[6]:7d}  (at {t[7] - base})")
    print(f"epilogue                     +{t[26] - t[7]:7d}  (total {t[26] - base})")
    sys.exit(0)
for r in range(1, 7):
    a, b = t[8 + 2 * (r - 1)], t[9 + 2 * (r - 1)]
    nxt = t[8 + 2 * r] if r < 6 else t[6]
    print(f"  round {r}: G1(r+1) + half act {b - a:6d}   G2(r-1) + half act + barrier {nxt - b:6d}")
print(f"last round + G2(nr-1)        +{t[7] - t[6]:7d}  (at {t[7] - base})")
print(f"epilogue                     +{t[26] - t[7]:7d}  (total {t[26] - base})")
