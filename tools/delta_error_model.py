#!/usr/bin/env python3
"""Why the bf16 attention backward takes delta = rowsum(dO . O) from O + its rounding residual (include/cvft.h `o_lo`).
CPU model (float64 with bf16 rounding where the kernels round): one head, T = 500 keys, d = 64.  dS = P (dP - delta) is rounded
to bf16 for the dQ / dK products as in csrc/attn_mfma32.hip; three sources of delta are compared against the exact gradients:
  (a) the bf16-ROUNDED forward output (what round 3 shipped),
  (b) the forward's fp32 output, i.e. bf16 O + bf16 residual (what ships now),
  (c) sum_j P_ij dP_ij from the backward's own recomputed P (a second pass over the keys, ~+50 % of the backward).
When the value rows share a large common component (dP nearly constant over the keys: the estimator's mid blocks, softmax nearly
flat) the error of (a) is a COMMON-MODE error of every dS in the row and dominates dQ / dK; (b) removes it at the cost of one more
8-byte store per lane in the forward and one more fragment load in the backward.   usage: python tools/delta_error_model.py"""
import torch

torch.manual_seed(0)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


T, d, sc = 500, 64, 0.125
for vmean, vnoise, qscale in ((1.0, 1.0, 0.3), (3.0, 0.2, 0.3), (3.0, 0.05, 0.3), (1.0, 1.0, 3.0)):
    q = bf(torch.randn(T, d, dtype=torch.float64) * qscale)
    k = bf(torch.randn(T, d, dtype=torch.float64) * qscale)
    v = bf(torch.randn(T, d, dtype=torch.float64) * vnoise + torch.randn(1, d, dtype=torch.float64) * vmean)
    do = bf(torch.randn(T, d, dtype=torch.float64))
    s = q @ k.T * sc
    p = torch.softmax(s, -1)
    m = s.max(-1, keepdim=True).values
    pb = bf(torch.exp(s - m))                      # the forward packs exp(s - m) to bf16 for the PV product ...
    l = pb.sum(-1, keepdim=True)                   # ... and sums the PACKED values for the denominator
    O = (pb @ v) / l
    pk = torch.exp(s - (m + torch.log(l)))         # the backward's P = exp(s - lse)
    dP = do @ v.T
    ref_delta = (p * dP).sum(-1, keepdim=True)
    dS_ref = p * (dP - ref_delta) * sc
    dq_ref, dk_ref = dS_ref @ k, dS_ref.T @ q
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"V = {vmean} x common + {vnoise} x noise, q/k scale {qscale}: max prob {float(p.max()):.3f}, |dP| {float(dP.abs().mean()):.2f}, "
          f"|dP - delta| {float((dP - ref_delta).abs().mean()):.3f}")
    for name, dl in (("(a) bf16 O", (do * bf(O)).sum(-1, keepdim=True)), ("(b) O + residual", (do * (bf(O) + bf(O - bf(O)))).sum(-1, keepdim=True)),
                     ("(c) sum_j P dP", (pk * dP).sum(-1, keepdim=True))):
        dS = bf(pk * (dP - dl) * sc)
        print(f"    delta from {name:18s}: |delta error| {float((dl - ref_delta).abs().mean()):.2e}   dQ rel-L2 {rel(dS @ k, dq_ref):.2e}   dK rel-L2 {rel(dS.T @ q, dk_ref):.2e}")
