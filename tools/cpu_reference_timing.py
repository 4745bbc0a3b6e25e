#!/usr/bin/env python3
"""CPU timing of the REFERENCE's own modules beside the oracle port (BASELINE.md section 2 protocol) -- build container only
(needs /root/reference; same assembly as tools/make_golden.py: flow_model.build_flow_model + vendored ConformerEncoder /
InterpolateRegulator, cosyvoice.llm.llm.TransformerLM behind two empty import shims, lora.apply_lora_to_model,
llm_flow_model.JointLLMFlowModel; train_joint.py itself needs pytorch_lightning, which is not installed, so the loop is the
hand-rolled AdamW + LambdaLR + clip loop of the same arithmetic, train_joint.py:198-226, 349-360).

fp32, train() mode, torch.set_num_threads(<all cores>), synthetic (text, 80 x 500 mel) utterances, 3 warm-up + >= 5 timed optimiser
steps; utterances/s = utterances processed / wall time (forward + backward + optimiser).  The oracle port (oracle/ref_math.py,
what bench.py's cpu_baseline times on the GPU box's host) runs on the same batches for comparison.

usage: python tools/cpu_reference_timing.py [--batch 2] [--steps 5] [--out profiles/r3_cpu_reference_timing.json]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G          # imports the reference's modules (build container only)
import torch
import torch.nn as nn

from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--out", default=os.path.join(G.REPO, "profiles", "r3_cpu_reference_timing.json"))
    a = ap.parse_args()
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    flow = G.build_ref_flow('vendored')
    llm = G.build_ref_llm(G.FULL_LLM)
    G.ref_lora.apply_lora_to_model(flow, r=16, lora_alpha=32, lora_dropout=0.05, target_modules=G.FLOW_TARGETS)
    G.ref_lora.apply_lora_to_model(llm, r=16, lora_alpha=32, lora_dropout=0.15, target_modules=G.LLM_TARGETS)
    jm = G.ref_joint.JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
    jm.train()
    params = [p for p in jm.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01, betas=(0.9, 0.999))
    total = a.warmup + a.steps
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, (s + 1) / 2) * 0.5 * (1 + math.cos(math.pi * s / max(1, total))))
    batches = [synth_batch([a.frames] * a.batch, seed=1234 + i) for i in range(total)]
    dev = torch.device('cpu')

    def ref_step(b):
        out = jm(b, dev)
        out['loss'].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        sched.step()
        opt.zero_grad(set_to_none=True)
        return float(out['loss'])

    print(f"[cpu_reference_timing] reference modules, joint, B={a.batch}, T={a.frames}, r=16, fp32, train(), {cores} threads", flush=True)
    for i in range(a.warmup):
        t0 = time.time()
        l = ref_step(batches[i])
        print(f"  warm-up {i}: {time.time() - t0:.2f} s  loss {l:.4f}", flush=True)
    t0 = time.time()
    for i in range(a.steps):
        ref_step(batches[a.warmup + i])
    ref_s = (time.time() - t0) / a.steps
    print(f"  reference: {ref_s:.2f} s/step = {a.batch / ref_s:.3f} utt/s", flush=True)

    # the oracle port on the same batches (eval-mode math like bench.py's cpu_baseline: forward + backward, no optimiser)
    from oracle import ref_math as R
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws
    sd_f = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    sd_l = {k: v.detach().clone() for k, v in llm.state_dict().items()}
    for sd in (sd_f, sd_l):
        for k, v in sd.items():
            if 'lora_' in k:
                v.requires_grad_(True)
    cfg = R.OracleConfig(flow_lora_scale=2.0, llm_lora_scale=2.0)

    def port_step(b, i):
        out = R.joint_forward(sd_l, sd_f, b, cfm_draws(a.batch, a.frames, i), cfg, 'joint', 2.0, 1.0)
        ps = [v for sd in (sd_f, sd_l) for v in sd.values() if v.requires_grad]
        torch.autograd.grad(out['loss'], ps, allow_unused=True)
    port_step(batches[0], 0)
    t0 = time.time()
    n = max(2, a.steps // 2)
    for i in range(n):
        port_step(batches[a.warmup + i], i)
    port_s = (time.time() - t0) / n
    print(f"  oracle port: {port_s:.2f} s/step = {a.batch / port_s:.3f} utt/s", flush=True)
    rec = dict(what="CPU timing of the reference's own modules (BASELINE.md section 2) beside the oracle port", cores=cores,
               cpu=open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t') if os.path.exists('/proc/cpuinfo') else '?',
               torch=torch.__version__, batch=a.batch, frames=a.frames, lora_r=16, dtype="f32", mode="joint, train() (dropout on)",
               warmup_steps=a.warmup, timed_steps=a.steps,
               reference=dict(s_per_step=ref_s, utt_per_s=a.batch / ref_s, includes="forward + backward + clip + AdamW + LambdaLR"),
               oracle_port=dict(s_per_step=port_s, utt_per_s=a.batch / port_s, includes="forward + backward (eval-mode math), no optimiser",
                                steps=n))
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
