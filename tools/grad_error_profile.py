#!/usr/bin/env python3
"""Where does the bf16 path's per-tensor LoRA-gradient error come from?  Runs the full-size flow branch (CosyVoice-300M dims,
ragged B = 2, T and r from argv) in fp32 (the path pinned to the reference at 8e-4) and in bf16 on the same batch / draws and
prints, per adapter tensor in network order, the relative L2 error of the bf16 gradient and the tensor's share of the total
gradient norm.  VERDICT round 2: "a 14 % error on an early-layer adapter gradient at T = 1000 / r = 64 is itself unexplained".
usage: grad_error_profile.py [T=1000] [r=64]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import build_flow_product, lora_grads, rel
from cosyvoice_lora_finetune_framework_amd.modules import Numerics
from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
DEV = "cuda"
targets = ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'w_1', 'w_2']
meta = dict(lora=dict(r=r, alpha=2 * r, targets=targets), weight_seed=1)
lens = [T, int(T * 0.874)]
batch = synth_batch(lens, text_lens=[40, 33], seed=1234)
draws = cfm_draws(2, T, 4321)
res = {}
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
    m = build_flow_product(meta, DEV, Numerics(dtype=dt))
    out = m.forward_no_prompt(batch, DEV, draws)
    out["loss"].backward()
    res[name] = (float(out["loss"]), {k: v.detach().float().cpu() for k, v in lora_grads(m).items()})
    del m
    torch.cuda.empty_cache()
g32, g16 = res["fp32"][1], res["bf16"][1]
tot = sum(float(v.double().pow(2).sum()) for v in g32.values()) ** 0.5
print(f"T={T} r={r}: loss fp32 {res['fp32'][0]:.6f} bf16 {res['bf16'][0]:.6f}; total grad norm {tot:.4e}")
print(f"{'tensor':78s} {'rel-L2 err':>10s} {'norm share':>10s} {'|g|':>10s}")
rows = []
for k in g32:
    n = float(g32[k].norm())
    rows.append((k, rel(g16[k], g32[k]), n / tot, n))
for k, e, s, n in rows:
    flag = " <--" if e > 0.1 else ""
    print(f"{k:78s} {e:10.3e} {s:10.3e} {n:10.3e}{flag}")
big = [x for x in rows if x[2] > 0.02]
print(f"\nworst error among tensors holding > 2 % of the gradient norm: {max(x[1] for x in big):.3e}; among all: {max(x[1] for x in rows):.3e}")
num = sum(float((g16[k].double() - g32[k].double()).pow(2).sum()) for k in g32) ** 0.5
print(f"whole-gradient rel-L2 error: {num / tot:.3e}")
