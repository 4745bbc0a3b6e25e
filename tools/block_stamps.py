#!/usr/bin/env python3
"""Phase stamps of the fused block-tail forward kernel (diagnostic build -DBF_STAMPS -> libcvft_bfstamps.so; never the
product library): where wave 0 of one workgroup spends its cycles.  Usage (GPU box):
    bash tools/build_block_stamps.sh && CVFT_LIB_PATH=.../libcvft_bfstamps.so python tools/block_stamps.py"""
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
blk = Mo.BasicTransformerBlock(256, 8, 64, 0.0, "gelu").to(dev)
for p in blk.parameters():
    p.requires_grad_(False)
M = int(os.environ.get("M", 4000))
o = torch.randn(M, 512, device=dev, dtype=dt)
x0 = torch.randn(M, 256, device=dev, dtype=dt)
for _ in range(50):
    with torch.no_grad():
        blk._tail(o, x0, "gelu_erf")
torch.cuda.synchronize()
lib = C.CDLL(cb.LIB_PATH)
buf = (C.c_ulonglong * 32)()
assert lib.cvft_debug_block_stamps(buf) == 0
t = list(buf)
names = {0: "start", 1: "out-proj MFMAs done", 2: "reduce done", 3: "x1 + residual done", 4: "LN + y tile + frags", 5: "first W1 product",
         6: "loop done", 7: "final reduce done"}
base = t[0]
for i in (1, 2, 3, 4, 5):
    print(f"{names[i]:28s} +{t[i] - t[i - 1]:7d} cycles  (at {t[i] - base})")
for k in range(8):
    nxt = t[9 + k] if k < 7 else t[6]
    print(f"  tile {k}: {nxt - t[8 + k]:7d} cycles")
print(f"{names[6]:28s} at {t[6] - base}")
print(f"{names[7]:28s} +{t[7] - t[6]:7d} cycles  (at {t[7] - base})")
