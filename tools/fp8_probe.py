#!/usr/bin/env python3
"""Exact-data probe of the block-scaled fp8 MFMA's operand layout (csrc/gemm_fp8.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
g = torch.Generator().manual_seed(0)
vals = torch.tensor([-4., -3., -2., -1.5, -1., -.5, 0., .5, 1., 1.5, 2., 3., 4., 6., 8., .25])
A = vals[torch.randint(0, 16, (16, 128), generator=g)]
B = vals[torch.randint(0, 16, (16, 128), generator=g)]
a8, b8 = A.to(torch.float8_e4m3fn), B.to(torch.float8_e4m3fn)
assert torch.equal(a8.float(), A) and torch.equal(b8.float(), B)
ad, bd = a8.view(torch.uint8).cuda(), b8.view(torch.uint8).cuda()
C = torch.zeros(16, 16, device="cuda")
cb.check(cb.lib().cvft_debug_mfma_fp8_probe(ad.data_ptr(), bd.data_ptr(), C.data_ptr(), None), "probe")
torch.cuda.synchronize()
ref = A.double() @ B.double().t()
err = (C.cpu().double() - ref).abs().max().item()
print("max abs err", err, "ref range", ref.abs().max().item())
print("OK" if err == 0 else "MISMATCH")
