#!/usr/bin/env python3
"""Phase stamps of the 64-row qkv forward kernel (csrc/block_qkv_wide.hip; diagnostic build, never the product library):
    bash tools/build_block_stamps.sh block_qkv_wide && CVFT_LIB_PATH=.../libcvft_bfstamps.so python tools/block_stamps_qkv.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("P", "0.05")
import torch
import tools.bench_block_qkv as B
from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb

B.timeit = lambda f: [f() for _ in range(30)] and torch.cuda.synchronize() or 1.0
_orig = B.main


def main():
    import builtins
    B.main.__globals__["print"] = lambda *a, **k: None
    _orig()
    lib = C.CDLL(cb.LIB_PATH)
    buf = (C.c_ulonglong * 32)()
    assert lib.cvft_debug_block_stamps(buf) == 0
    t = list(buf)
    names = ["x tile + params in LDS", "LayerNorm -> y tiles", "rank-side MFMAs + copies", "exchange + U", ]
    for i, n in enumerate(names):
        builtins.print(f"{n:28s} +{t[i + 1] - t[i]:7d}  (at {t[i + 1] - t[0]})")
    for i in range(6):
        nxt = t[5 + 2 * (i + 1)] if i < 5 else t[17]
        builtins.print(f"  tile {i}: MFMAs {t[6 + 2 * i] - t[5 + 2 * i]:6d}   ext + staging/store {nxt - t[6 + 2 * i]:6d}")
    builtins.print(f"total {t[17] - t[0]}")
    builtins.print("backward:")
    builtins.print(f"dY chunk 0, x, dres in LDS   +{t[21] - t[20]:7d}")
    for q in range(6):
        builtins.print(f"  chunk {q}: {t[23 + q] - t[22 + q]:6d}")
    builtins.print(f"V exchange                   +{t[29] - t[28]:7d}")
    builtins.print(f"side term                    +{t[30] - t[29]:7d}")
    builtins.print(f"LN backward: sums {t[12] - t[30]}, exchange {t[13] - t[12]}, dx -> tile {t[14] - t[13]}, barrier {t[15] - t[14]}, store {t[31] - t[15]}  (total {t[31] - t[20]})")


if __name__ == "__main__":
    main()
