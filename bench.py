#!/usr/bin/env python3
"""bench.py -- train utterances/sec of the joint LLM+Flow LoRA step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  W untimed warm-up steps, EXACTLY K timed steps bracketed by
barrier + torch.cuda.synchronize(), MAX over ranks, rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one per-GPU batch of synthetic utterances
(500-frame x 80-mel clips, 290 speech tokens, 40 text tokens; random-init CosyVoice-300M dims,
LoRA r=16 alpha=32): forward + backward + LoRA-gradient all-reduce + clip + AdamW.  Weak
scaling: per-GPU batch fixed (default 16 = BASELINE configs[2] at N=1, configs[3] at N=8).
By default the steps are the product loop's: train_joint.Trainer.fit over W + K host batches (per step: index maps,
host -> device copies, replay of the trainer's captured micro-step hipGraph, all-reduce, clip + AdamW);
--via-trainer 0 replays one pre-staged batch instead (within 2 % of each other on one MI355X).

Extra objects on the JSON line:
  roofline     -- dominant kernel (by time) of an event-instrumented step run right after the timed
                  region, same process / stream / data: algorithmic FLOPs per launch / mean launch
                  duration vs the dense bf16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline -- the CPU oracle (torch fp32, all host cores; oracle/ref_math.py = port of the
                  reference's math) on a bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / shared device tensors fail with hipIpcGetMemHandle otherwise);
# the driver's environment exports it already -- this only covers a launch from a shell that lost it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md chip table)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0      # HBM3E (same table)


# algorithmic work per utterance, forward + backward, 2 x MAC (torch flop counter on the oracle, LoRA r = 16)
GFLOP_PER_UTT = {500: {"joint": 468.9, "flow_only": 154.08, "llm_only": 314.83},
                 1000: {"joint": 1015.0, "flow_only": 386.08, "llm_only": 628.94}}


def step_record(workload, frames, utt_per_s_per_gpu, dtype):
    """what north_star grades: the whole step's share of the dense MFMA peak of ONE GPU"""
    gf = GFLOP_PER_UTT.get(frames, {}).get(workload)
    if gf is None:
        return None
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    tf = utt_per_s_per_gpu * gf / 1e3
    return {"tflops": tf, "frac_of_mfma_peak": tf / peak, "gflop_per_utt": gf, "peak_tflops": peak}


def _build_id():
    from cosyvoice_lora_finetune_framework_amd.build_id import csrc_sha16
    return csrc_sha16()


def _stale(d, path):
    """None when the profile record `d` was written from the kernel sources this process runs (its `csrc_sha16` equals
    build_id.csrc_sha16()); else the reason the bench line prints instead of the figure."""
    have, want = d.get("csrc_sha16"), _build_id()
    if have == want:
        return None
    return (f"{os.path.join('profiles', os.path.basename(path))} was recorded on kernel sources {have or '(no identity recorded)'}, "
            f"this build is {want}: rerun tools/refresh_profiles.sh")


def trace_summary(workload, batch, frames):
    """launches and kernel time per step of this configuration from the committed rocprofv3 kernel trace of the same command
    (profiles/*step_summary.json, written by tools/prof_summary.py; the trace cannot be taken inside the bench process)"""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*step_summary*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("batch") == batch and d.get("frames") == frames:
            why = _stale(d, f)
            if why:
                return {"launches_per_step": None, "kernel_ms_per_step": None, "source": None, "not_quoted": why}
            return {"launches_per_step": d["launches_per_step"], "kernel_ms_per_step": d["kernel_ms_per_step"],
                    "source": os.path.join("profiles", os.path.basename(f))}
    return None


def trace_kernel_time(workload, batch, frames):
    """{bracket label: chip-equivalent ms per step} of this configuration's REAL step -- three chains on three streams -- from the
    committed kernel trace of this build (profiles/*cu_time.json, tools/cu_time.py: kernel time x the share of the 256 CUs a launch
    holds -- the quantity the step's time follows), or None (no record, or made on other kernel sources).
    Used only to CHOOSE which kernel the roofline record is about: the instrumented step that measures it runs the chains one
    after the other on one stream, where the chip-filling GEMMs are faster than in the step and the row-tile chain kernels are not."""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*cu_time*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("batch") == batch and d.get("frames") == frames:
            if _stale(d, f) or "kernels" not in d:
                return None
            out = {}
            for k in d["kernels"]:
                n = k["kernel"]
                m = re.search(r"gemm_glds_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi0ELi(\d+)E", n)
                if m:
                    lab = "gemm_glds_kernel<bf16,%s,%s,%s,%s,ns%s,regepi>" % m.groups()
                elif "gemm_p256_kernel" in n:
                    lab = "gemm_p256_kernel<bf16,256,256,2,4,ring10>"
                else:
                    m = re.search(r"(block_(?:link|tail|qkv))(?:_wide8|_wide|_lean)?_(fwd|bwd)_kernel", n)
                    if not m:
                        continue
                    lab = m.group(1) + "_" + m.group(2)
                out[lab] = out.get(lab, 0.0) + k["chip_ms_per_step"]
            return out
    return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota (the GPU box exposes
    all host CPUs through os.cpu_count() but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def build(workload, dtype, device, r, alpha, dropout=False):
    from cosyvoice_lora_finetune_framework_amd.flow_model import build_flow_model
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.llm_model import build_llm_model
    from cosyvoice_lora_finetune_framework_amd.lora import apply_lora_to_model
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.config import JOINT_TRAINING_CONFIG as JC
    num = Numerics(dtype=dtype)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        flow = build_flow_model(None, 'cpu', numerics=num)
        llm = build_llm_model(None, 'cpu', numerics=num) if workload != 'flow_only' else torch.nn.Identity()
        if workload in ('joint', 'flow_only'):
            apply_lora_to_model(flow, r=r, lora_alpha=alpha, lora_dropout=JC['flow_lora']['lora_dropout'] if dropout in (1, 2) else 0.0,
                                target_modules=JC['flow_lora']['target_modules'])
        if workload in ('joint', 'llm_only'):
            apply_lora_to_model(llm, r=r, lora_alpha=alpha, lora_dropout=JC['llm_lora']['lora_dropout'] if dropout in (1, 2) else 0.0,
                                target_modules=JC['llm_lora']['target_modules'])
        if workload == 'llm_only':
            flow.requires_grad_(False)
        jm = JointLLMFlowModel(llm, flow, workload, llm_loss_weight=JC['llm_loss_weight'],
                               flow_loss_weight=JC['flow_loss_weight'])
    jm = jm.to(device)
    # dropout 1 (bench default) = the reference's training regularisation (LoRA dropout 0.15 / 0.05, encoder dropouts 0.1), i.e.
    # the step as trainer.fit runs it; 0 = eval(), dropout off, like the parity fixtures and the CPU baseline
    if dropout == 2:          # LoRA dropout only (diagnostic): encoder dropouts off
        for m in jm.modules():
            if hasattr(m, 'dropout_rate'):
                m.dropout_rate = 0.0
            if hasattr(m, 'embed') and hasattr(m.embed, 'out'):
                m.embed.out[2].p = 0.0
    return jm.train() if dropout else jm.eval()


def cpu_baseline(jm, workload, T, seconds_budget=30.0):
    """Time the CPU oracle (port of the reference math) on a bounded sample: 4 utterances per step (the GPU's batch of
    16 would take ~15 s per CPU step: over the ~30 s budget; utterances/s is flat in B on the CPU between 1 and 4)."""
    from oracle import ref_math as R
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"[bench] cpu_baseline: timing the CPU oracle on {cores} host threads ...")
    sd_f = {k: v.detach().float().cpu() for k, v in jm.flow.state_dict().items()}
    sd_l = {k: v.detach().float().cpu() for k, v in jm.llm.state_dict().items()} if workload != 'flow_only' else {}
    for sd in (sd_f, sd_l):
        for k, v in sd.items():
            if 'lora_' in k:
                v.requires_grad_(True)
    cfg = R.OracleConfig(flow_lora_scale=2.0, llm_lora_scale=2.0)
    B = 4
    batch = synth_batch([T] * B, seed=99)
    draws = cfm_draws(B, T, 5)

    def one():
        out = R.joint_forward(sd_l, sd_f, batch, draws, cfg, workload, jm.llm_loss_weight, jm.flow_loss_weight)
        params = [v for sd in (sd_f, sd_l) for v in sd.values() if v.requires_grad]
        torch.autograd.grad(out['loss'], params, allow_unused=True)
        return float(out['loss'])
    t0 = time.time()
    one()
    first = time.time() - t0
    log(f"[bench] cpu_baseline: warm-up step took {first:.1f} s")
    n = max(1, min(5, int((seconds_budget - first) / max(first, 1e-3))))
    t0 = time.time()
    for _ in range(n):
        one()
    dt_ = (time.time() - t0) / n
    return {"value": B / dt_, "unit": "utterances/s", "cores": cores, "kind": "port", "reference_modules": reference_modules_record(),
            "sample": f"{n} fwd+bwd steps of {B} utterance(s) ({T}-frame mel, {workload}), torch fp32 CPU oracle "
                      f"(oracle/ref_math.py), same random-init weights, no dropout; optimiser step excluded (negligible)"}


def reference_modules_record():
    """The reference's OWN modules timed on the CPU (it cannot travel to the GPU box): the newest committed
    profiles/r*_cpu_reference_timing.json (tools/cpu_reference_timing.py, build container), quoted beside the port's figure with
    its own host and core count -- a different host than `cpu_baseline.value`'s, so context, not a ratio."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*cpu_reference_timing.json")), reverse=True):
        try:
            d = json.load(open(f))
            return {"value": d["reference"]["utt_per_s"], "unit": "utterances/s", "cores": d["cores"], "cpu": d.get("cpu"),
                    "s_per_step": d["reference"]["s_per_step"], "batch": d["batch"], "frames": d["frames"], "mode": d.get("mode"),
                    "includes": d["reference"].get("includes"), "port_on_that_host_utt_per_s": d.get("oracle_port", {}).get("utt_per_s"),
                    "source": os.path.join("profiles", os.path.basename(f)),
                    "note": "measured in the build container, not on this run's host"}
        except Exception:
            continue
    return None


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed PMC run of this same command (rocprofv3 --pmc cannot
    run inside the bench process): profiles/*pmc_traffic.json, written by tools/pmc_traffic.py.  None when the file's
    kernel is not the one that dominates this run."""
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel") == kernel:
            why = _stale(d, f)
            if why:
                return None, "not quoted: " + why
            return d["traffic_bytes_per_launch"], os.path.join("profiles", os.path.basename(f))
    return None, None


def pmc_mfma_busy(kernel: str):
    """MFMA-busy fraction of `kernel` (share of the chip's matrix-pipe cycles a launch keeps busy) from the committed
    counter run of the same step: profiles/*counters.json, written by tools/counters.py (SQ_VALU_MFMA_BUSY_CYCLES /
    (1024 SIMDs x GRBM_GUI_ACTIVE / 8))."""
    import glob
    import re
    m = re.match(r"gemm_glds_kernel<bf16,(\d+),(\d+),(\d+),(\d+),ns(\d+)", kernel)
    if m:
        bm, bn, wm, wn, ns = m.groups()
        key = f"gemm_glds<{bm}ELi{bn}ELi{wm}ELi{wn}ELi0ELi{ns}E"
    elif kernel.startswith("gemm_p256_kernel"):
        key = "gemm_p256"
    elif kernel.startswith("block_"):                 # the estimator's chain kernels: bracket name + "_kernel<...>"
        key = kernel + "_kernel"
    else:
        return None, None
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*counters.json")), reverse=True):
        try:
            doc = json.load(open(f))
            ks = doc["kernels"]
        except Exception:
            continue
        hit = [v for k, v in ks.items() if k.startswith(key)]
        if hit:
            why = _stale(doc, f)
            if why:
                return None, "not quoted: " + why
            n = sum(v["launches_sampled"] for v in hit)
            return sum(v["mfma_busy"] * v["launches_sampled"] for v in hit) / n, os.path.join("profiles", os.path.basename(f))
    return None, None


def trainer_run(a, jm, workload, dev, dtype, rank, world, steps, warmup):
    """The product loop: train_joint.Trainer.fit over W + K fresh synthetic batches; the timed region is bracketed from the
    trainer's per-step hook (barrier + synchronize on both sides).  Returns (seconds of the K steps, MAX over ranks; loss;
    graph statistics; the trainer)."""
    from cosyvoice_lora_finetune_framework_amd import dp
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, SyntheticLoader, Trainer
    module = JointLightningModule(workload, learning_rate=2e-4, min_lr=1e-6, warmup_steps=10, weight_decay=0.01,
                                  model=jm, numerics=Numerics(dtype=dtype))
    # per-rank host budget (dp.host_budget, applied by Trainer.fit when world > 1): cores // world per rank, the replay thread and
    # the prefetch thread on cores of their own, torch's intra-op pool on the rest -- eight ranks do not oversubscribe one host
    if world == 1:
        torch.set_num_threads(max(1, min(host_cores(), 16)))
    B, T = a.batch, a.frames
    loader = SyntheticLoader(warmup + steps, B, T, seed=1234, ragged=a.ragged, rank=rank, cache=True)
    marks = {}

    def hook(tr):
        if tr.global_step == warmup:
            dp.barrier()
            torch.cuda.synchronize()
            marks["t0"] = time.perf_counter()
        elif tr.global_step == warmup + steps:
            torch.cuda.synchronize()
            dp.barrier()
            torch.cuda.synchronize()
            marks["t1"] = time.perf_counter()

    tr = Trainer(max_epochs=1, accumulate_grad_batches=1, gradient_clip_val=1.0, save_checkpoints=False,
                 log_every_n_steps=0, train_mode=bool(a.dropout), use_graph=bool(a.graph), on_step_end=hook)
    log(f"[bench] rank {rank}: Trainer.fit [{workload}] ({'hipGraph micro-step' if a.graph else 'eager'}); warm-up {warmup}, timing {steps} steps")
    with contextlib.redirect_stdout(sys.stderr):
        tr.fit(module, loader)
    from cosyvoice_lora_finetune_framework_amd import train_joint as _tj
    if _tj._TIMING:
        torch.cuda.synchronize()
        TL = _tj._TIMING[-10:]
        cp = sum(e[0].elapsed_time(e[1]) for e in TL) / len(TL)
        gr = sum(e[1].elapsed_time(e[2]) for e in TL) / len(TL)
        gap = sum(TL[i][2].elapsed_time(TL[i + 1][0]) for i in range(len(TL) - 1)) / (len(TL) - 1)
        hc = sum(e[3][1] - e[3][0] for e in TL) / len(TL) * 1e3
        hr = sum(e[3][2] - e[3][1] for e in TL) / len(TL) * 1e3
        hg = sum(TL[i + 1][3][0] - TL[i][3][2] for i in range(len(TL) - 1)) / (len(TL) - 1) * 1e3
        log(f"[bench] trainer host timeline: copies {hc:.2f} ms, replay call {hr:.2f} ms, replay return -> next step's copies {hg:.2f} ms")
        log(f"[bench] trainer GPU timeline: static copies {cp:.3f} ms, graph {gr:.3f} ms, graph end -> next step's copies {gap:.3f} ms")
        _tj._TIMING.clear()
    el = torch.tensor([marks["t1"] - marks["t0"]], device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    return float(el), float(tr.callback_metrics.get("train_loss_epoch", float("nan"))), dict(tr.graph_stats), tr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="joint", choices=["joint", "flow_only", "llm_only"])
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--rank-lora", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=1, help="capture fwd+bwd in a hipGraph (0 = eager launches)")
    ap.add_argument("--dropout", type=int, default=1, help="1 (default) = train() mode with all of the reference's training dropouts (LoRA 0.15 / 0.05, encoder 0.1), as trainer.fit runs the step; 0 = eval() mode, dropout off, like the parity fixtures; 2 / 3 = LoRA / encoder dropouts only (diagnostic)")
    ap.add_argument("--fp8", type=int, default=0, help="1 = BASELINE configs[4] arithmetic: the frozen-W GEMMs of the LLM-sized linears in OCP e4m3 (per-token / per-channel scales); LoRA path, reductions and everything else stay bf16 / fp32")
    ap.add_argument("--via-trainer", type=int, default=1, help="1 (default) = time the steps inside train_joint.Trainer.fit (fresh host batch every step: index maps + H2D copies + the trainer's captured micro-step graph + all-reduce + clip + AdamW) ; 0 = replay one pre-staged batch (no trainer, no per-step host work)")
    ap.add_argument("--ragged", type=int, default=0, help="--via-trainer: 1 = utterance lengths uniform in [0.6 T, T], one utterance keeps T (one batch layout); 2 = none pinned: every batch has its own T_max / Lt_max and the trainer fits it to a captured layout (SHAPE_SLACK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-branches", action="store_true", help="skip the llm_only / flow_only legs of roofline.branches (joint, N = 1)")
    a = ap.parse_args()

    from cosyvoice_lora_finetune_framework_amd import dp
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch

    HF.FP8_ON = bool(a.fp8)
    rank, local, world = dp.init_from_env()
    if world != a.gpus:
        log(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    jm = build(a.workload, dtype, dev, a.rank_lora, 2 * a.rank_lora, a.dropout)
    opt = None if a.via_trainer else FlatAdamW([p for p in jm.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01,
                                              max_grad_norm=1.0)
    B, T = a.batch, a.frames
    batch = jm.prepare_batch(synth_batch([T] * B, seed=1234 + rank), dev)

    def fwd_bwd():
        out = jm(batch, dev)
        with HF.LoraGradSink():
            out['loss'].backward()
        return out['loss'].detach()

    graph = None
    static_loss = None
    trainer_stats = None
    if a.via_trainer:
        if a.warmup < 1:
            log("[bench] --via-trainer captures the batch layout in its first step: raising --warmup to 1")
            a.warmup = 1
        elapsed, final_loss, trainer_stats, tr = trainer_run(a, jm, a.workload, dev, dtype, rank, world, a.steps, a.warmup)
        opt = tr.optimizer
        graph = True if (a.graph and trainer_stats["replays"] > 0) else None
    elif a.graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):                 # allocator / pack warm-up on the side stream
                    fwd_bwd()
                    opt.zero_grad()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (the RCCL watchdog thread keeps running)
                static_loss = fwd_bwd()
            opt.zero_grad()
        except Exception as e:                      # still the HIP path, just launched eagerly
            log(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager launches")
            graph = None
            torch.cuda.synchronize()
            opt.zero_grad()

    def step():
        if graph is not None:
            graph.replay()
            loss = static_loss
        else:
            loss = fwd_bwd()
        gscale = dp.allreduce_flat_grads(opt.flat_g)
        opt.step(gscale)
        opt.zero_grad()
        return loss

    if not a.via_trainer:
        log(f"[bench] rank {rank}: model ready ({'hipGraph' if graph is not None else 'eager'}); warm-up {a.warmup}, timing {a.steps} steps")
        for _ in range(a.warmup):
            step()
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], device=dev)
        if world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(el)
        final_loss = float(loss)

    log(f"[bench] timed region done: {elapsed / a.steps * 1e3:.2f} ms/step; device memory reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")
    if os.environ.get("CVFT_BENCH_HOSTLAUNCH") and not a.via_trainer and graph is not None:
        hs, tot = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            hs.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
        log(f"[bench] idle-GPU replay: host call {min(hs):.2f} ms, to completion {min(tot):.2f} ms")
    roof = None
    cabi_calls = None
    if rank == 0 and not a.no_roofline:              # (the instrumented step has no collective in it: rank 0 alone runs it)
        # event-instrumented eager step: every tap-GEMM launch bracketed by HIP events on the launch stream
        # An event pair costs time of its own (two timestamp packets on the queue): the empty bracket is measured here and
        # taken off every record, else the 8-us launches of this step read 50 % long against the rocprofv3 kernel trace.
        ov = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            ov.append((e0, e1))
        torch.cuda.synchronize()
        evt_ms = sorted(x.elapsed_time(y) for x, y in ov)[len(ov) // 2]
        # The eager step is host-bound (the host needs ~2.5x the GPU's time to enqueue it), so an event bracket would also
        # time the host's gap between `record` and the launch call.  The instrumented step therefore runs BEHIND a spin
        # kernel that keeps the queues blocked while the host enqueues it: the GPU then executes the launches back to back
        # (three chains concurrently, as in the captured step) and the brackets time kernels, not the host.
        # The chains run one after the other on ONE stream here (BRANCH_STREAMS off): a bracket then holds exactly one
        # kernel, as in the rocprofv3 kernel trace (which serialises dispatches) -- concurrent chains stretch each other's
        # launches by 1.2-1.5x, which is a property of the step, not of the kernel the roofline is about.
        from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
        branch_streams, J.BRANCH_STREAMS = J.BRANCH_STREAMS, False
        J.CHAINS_HINT = HF.lib().cvft_concurrent_chains()      # same tile choices as the timed (multi-chain) step
        HF.PROFILE = []
        torch.cuda.synchronize()
        from cosyvoice_lora_finetune_framework_amd.hipops import binding as _cb
        c_before = _cb.CALLS
        h0 = time.perf_counter()
        fwd_bwd()                                   # pass 1: host enqueue time of the instrumented step (records dropped)
        host_ms = (time.perf_counter() - h0) * 1e3
        cabi_calls = _cb.CALLS - c_before           # libcvft entry-point calls of one forward + backward (one chain per branch)
        opt.zero_grad()
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        torch.cuda._sleep(20_000_000)
        c1.record()
        torch.cuda.synchronize()
        cyc_per_ms = 20_000_000 / max(c0.elapsed_time(c1), 1e-3)
        HF.PROFILE = []
        torch.cuda._sleep(int(min(1.3 * host_ms + 5.0, 400.0) * cyc_per_ms))
        fwd_bwd()
        opt.zero_grad()
        torch.cuda.synchronize()
        J.BRANCH_STREAMS = branch_streams
        J.CHAINS_HINT = 0
        groups = {}
        for rec in HF.PROFILE:
            # the masked-extension instantiation (",xdrop": lora_dropout dgrad) is the same kernel with a different rank tail
            g = groups.setdefault(rec["kernel"].replace(",xdrop", ""), {"ms": 0.0, "flop": 0.0, "n": 0, "bytes": 0.0})
            g["ms"] += max(rec["start"].elapsed_time(rec["end"]) - evt_ms, 1e-4)
            g["flop"] += rec["flop"]
            g["bytes"] += rec.get("bytes", 0.0)
            g["n"] += 1
        if os.environ.get("CVFT_BENCH_DUMP_PROFILE"):      # diagnostic: per-(kernel, shape, epilogue) table of the instrumented step
            per = {}
            for rec in HF.PROFILE:
                key = f"{rec['kernel']} {rec.get('shape')} {rec.get('epi', '')}"
                g = per.setdefault(key, {"ms": 0.0, "flop": 0.0, "n": 0})
                g["ms"] += max(rec["start"].elapsed_time(rec["end"]) - evt_ms, 1e-4)
                g["flop"] += rec["flop"]
                g["n"] += 1
            rows = sorted(per.items(), key=lambda kv: -kv[1]["ms"])
            with open(os.environ["CVFT_BENCH_DUMP_PROFILE"], "w") as fh:
                for k, g in rows:
                    fh.write(f"{g['ms']:8.3f} ms  n {g['n']:4d}  avg {g['ms'] * 1e3 / g['n']:7.1f} us  {g['flop'] / (g['ms'] * 1e-3) / 1e12:7.1f} TF/s  {k}\n")
        HF.PROFILE = None
        if groups:
            mfma_peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
            ridge = mfma_peak * 1e12 / (PEAK_HBM_GBPS * 1e9)        # FLOP per byte where the two roofs meet

            def line(name, g):
                """roofline record of one kernel: the roof is HBM when its algorithmic intensity is below the ridge"""
                hbm = g["flop"] / max(g["bytes"], 1.0) < ridge
                ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9 if hbm else g["flop"] / (g["ms"] * 1e-3) / 1e12
                peak = PEAK_HBM_GBPS if hbm else mfma_peak
                traffic, traffic_src = pmc_traffic(name)
                busy, busy_src = pmc_mfma_busy(name)
                return {"bound": "hbm" if hbm else "mfma", "kernel": name, "achieved": ach, "peak": peak,
                        "unit": "GB/s" if hbm else "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                        "mfma_busy": busy, "mfma_busy_source": busy_src, "alg_bytes_per_launch": g["bytes"] / g["n"],
                        "launches_per_step": g["n"], "avg_launch_us": g["ms"] * 1e3 / g["n"],
                        "alg_gflop_per_launch": g["flop"] / g["n"] / 1e9, "intensity_flop_per_byte": g["flop"] / max(g["bytes"], 1.0)}
            # dominant = most CU-time in the REAL step (this build's committed trace); without one, most time in the instrumented step
            tk = trace_kernel_time(a.workload, B, T)
            cand = {k: v for k, v in (tk or {}).items() if k in groups}
            if cand:
                name = max(cand.items(), key=lambda kv: kv[1])[0]
                g = groups[name]
            else:
                name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
            roof = line(name, g)
            roof["chosen_by"] = ("chip-equivalent kernel time per step (time x share of the 256 CUs held) in this build's trace (profiles/*cu_time.json): " +
                                 ", ".join(f"{k} {v:.2f} ms" for k, v in sorted(cand.items(), key=lambda kv: -kv[1])[:4])) if cand \
                else "time in the instrumented single-stream step"
            roof["event_pair_overhead_us"] = evt_ms * 1e3
            roof["method"] = ("HIP events around every GEMM launch of one eager single-stream step enqueued behind a spin kernel "
                              f"(host enqueue {host_ms:.0f} ms hidden), empty event pair subtracted")
            mf = [(k, v) for k, v in groups.items() if v["flop"] / max(v["bytes"], 1.0) >= ridge and k != name]
            if mf:                                                                   # and the largest matrix-core-bound one
                roof["largest_mfma_bound"] = line(*max(mf, key=lambda kv: kv[1]["ms"]))
            roof["all_gemm_kernels"] = {k: {"ms": v["ms"], "tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12,
                                            "gbps": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "n": v["n"]}
                                        for k, v in groups.items() if not k.startswith("attn_")}
            # the fused attention launches of the same instrumented step (FLOPs of the visible query-key pairs only;
            # a rel-pos backward bracket holds its two launches, dQ and dK/dV)
            roof["attention_kernels"] = {k: {"ms": v["ms"], "tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12,
                                             "frac_of_mfma_peak": v["flop"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak,
                                             "avg_us": v["ms"] * 1e3 / v["n"], "n": v["n"]}
                                         for k, v in groups.items() if k.startswith("attn_")}

    # what north_star grades: whole-step share of the bf16 MFMA peak, per branch too (single-branch legs through the same trainer)
    if roof is not None:
        roof["step"] = step_record(a.workload, T, B * a.steps / elapsed, a.dtype)
        if roof["step"] is not None:
            roof["step"]["cabi_calls_per_step"] = cabi_calls
            ts = trace_summary(a.workload, B, T)
            roof["step"].update(ts or {"launches_per_step": None, "kernel_ms_per_step": None, "source": None})
        if a.workload == "joint" and world == 1 and a.via_trainer and not a.no_branches:
            roof["branches"] = {}
            for wl in ("llm_only", "flow_only"):
                try:
                    jb = build(wl, dtype, dev, a.rank_lora, 2 * a.rank_lora, a.dropout)
                    el_b, _, _, trb = trainer_run(a, jb, wl, dev, dtype, rank, world, 10, 3)
                    rec = step_record(wl, T, B * 10 / el_b, a.dtype) or {}
                    roof["branches"][wl] = dict(ms_per_step=el_b / 10 * 1e3, utt_per_s=B * 10 / el_b, **rec)
                    del jb, trb
                    torch.cuda.empty_cache()
                except Exception as e:
                    log(f"[bench] branch leg {wl} failed: {type(e).__name__}: {e}")
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(jm, a.workload, T)
        except Exception as e:
            log(f"[bench] cpu_baseline failed: {type(e).__name__}: {e}")

    if rank == 0:
        utt = world * B * a.steps
        line = {
            "metric": "train utterances/sec (500-frame mel, LoRA r=16) at 1/2/4/8 MI355X",
            "value": utt / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8-e4m3 frozen-W GEMMs + " + a.dtype) if a.fp8 else a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.workload} LLM+Flow LoRA r={a.rank_lora} step (fwd+bwd+allreduce+clip+AdamW), "
                                   f"CosyVoice-300M dims random-init, {T}-frame x 80-mel clips, "
                                   f"{int(T * 50 * 256 / 22050)} speech tokens, 40 text tokens",
                       "per_gpu_batch": B, "global_batch": world * B, "frames": T, "lora_r": a.rank_lora,
                       "parallelism": f"dp{world}", "launch": ("Trainer.fit + " if a.via_trainer else "") + ("hipGraph" if graph is not None else "eager"),
                       "dropout": bool(a.dropout), "final_loss": final_loss,
                       **({"trainer": trainer_stats, "ragged": a.ragged} if a.via_trainer else {})},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as _J
    if rank == 0 and _J.CHAIN_EVENTS:                 # CVFT_CHAIN_EVENTS=1 CVFT_CHAIN_BWD=1 (diagnostic): the LAST replay's chain ends
        for key, ms in sorted(_J.chain_event_offsets_ms().items(), key=lambda kv: kv[1]):
            print(f"[chain events] +{ms:7.3f} ms  {key}", file=sys.stderr, flush=True)
    if torch.distributed.is_initialized():
        dp.barrier()                                  # the other ranks wait for rank 0's roofline leg, then all leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
