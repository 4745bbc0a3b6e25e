"""BASELINE configs[0] (plumbing reference): 8 synthetic pairs, LoRA r=4, fp32, 2 epochs, batch 1,
accumulate 2 -- the product Trainer (flat AdamW + warmup-cosine + clip) on the HIP path must
reproduce the loss curve, learning rates, gradient norms and final LoRA tensors that the
REFERENCE's modules produced under torch.optim.AdamW / LambdaLR on the CPU
(tests/golden/train_tiny_log.json, tools/make_golden.py::gen_train)."""
import math

import pytest
import torch

from conftest import load_json, load_npz
from helpers import build_flow_product, build_llm_product, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_training_curve_matches_reference(tiny_meta):
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    log = load_json("train_tiny_log.json")
    hp = log["hp"]
    num = Numerics(dtype=torch.float32)
    flow = build_flow_product(tiny_meta["flow"], DEV, num)
    llm = build_llm_product(tiny_meta["llm"], DEV, num)
    jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
    module = JointLightningModule('joint', learning_rate=hp["lr"], min_lr=hp["min_lr"], warmup_steps=hp["warmup"],
                                  weight_decay=hp["wd"], model=jm, numerics=num)
    batches = [synth_batch([T], text_lens=[Lx], token_lens=[Lt], seed=100 + i, text_vocab=100, speech_vocab=50)
               for i, (T, Lx, Lt) in enumerate(log["lens"])]
    tr = Trainer(max_epochs=hp["epochs"], accumulate_grad_batches=hp["accum"], gradient_clip_val=hp["clip"], train_mode=False,
                 log_every_n_steps=1, save_checkpoints=False,
                 draws_fn=lambda ep, bi, b: cfm_draws(1, b["speech_feat"].shape[1], 1000 * ep + bi))
    tr.fit(module, batches)
    ref = [r for r in log["log"] if "lr" in r]
    assert len(tr.history) == len(ref) == log["total_steps"]
    for got, exp in zip(tr.history, ref):
        assert abs(got["lr"] - exp["lr"]) <= 1e-9 + 1e-6 * exp["lr"]
        for k in ("loss", "llm_loss", "flow_loss"):
            assert abs(got[k] - exp[k]) / abs(exp[k]) < 1e-4, (k, got, exp)     # north_star: curve to 1e-4
        assert abs(got["llm_acc"] - exp["llm_acc"]) < 1e-6
        assert abs(got["grad_norm"] - exp["grad_norm"]) / exp["grad_norm"] < 2e-3
    final = load_npz("train_tiny_final.npz")
    own = dict(jm.named_parameters())
    worst = max(rel(own[k], v) for k, v in final.items())
    assert worst < 1e-3, worst


def test_training_curve_bf16_tracks_reference(tiny_meta):
    """The arithmetic bench.py times (bf16 storage / MFMA operands, fp32 accumulation, fp32 LoRA masters and AdamW) on the same
    configs[0] run: 8 optimiser steps of the product Trainer against the REFERENCE's fp32 CPU curve (train_tiny_log.json), dropout
    off.  bf16 cannot meet the north_star's 1e-4 (that is the fp32 path's pin, above); what is pinned here is that the bf16 curve
    TRACKS the reference step for step -- per-step loss within BF16_CURVE_TOL, learning rates exact, gradient norms and the final
    LoRA tensors within their tolerances -- with every tolerance <= 3x the measured deviation (printed)."""
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    # relative, per step; measured on MI355X (this test prints them): loss 6.3e-4, llm_loss 8.5e-4, flow_loss 3.5e-4,
    # gradient norm 1.24e-3, final LoRA tensors 6.8e-3 -- tolerances <= 3x those
    BF16_CURVE_TOL = {"loss": 1.9e-3, "llm_loss": 2.5e-3, "flow_loss": 1.0e-3}
    BF16_GNORM_TOL, BF16_FINAL_TOL = 3.7e-3, 2.0e-2
    log = load_json("train_tiny_log.json")
    hp = log["hp"]
    num = Numerics(dtype=torch.bfloat16)
    flow = build_flow_product(tiny_meta["flow"], DEV, num)
    llm = build_llm_product(tiny_meta["llm"], DEV, num)
    jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
    module = JointLightningModule('joint', learning_rate=hp["lr"], min_lr=hp["min_lr"], warmup_steps=hp["warmup"],
                                  weight_decay=hp["wd"], model=jm, numerics=num)
    batches = [synth_batch([T], text_lens=[Lx], token_lens=[Lt], seed=100 + i, text_vocab=100, speech_vocab=50)
               for i, (T, Lx, Lt) in enumerate(log["lens"])]
    tr = Trainer(max_epochs=hp["epochs"], accumulate_grad_batches=hp["accum"], gradient_clip_val=hp["clip"], train_mode=False,
                 log_every_n_steps=1, save_checkpoints=False,
                 draws_fn=lambda ep, bi, b: cfm_draws(1, b["speech_feat"].shape[1], 1000 * ep + bi))
    tr.fit(module, batches)
    ref = [r for r in log["log"] if "lr" in r]
    assert len(tr.history) == len(ref) == log["total_steps"]
    worst = {k: 0.0 for k in BF16_CURVE_TOL}
    worst_g = 0.0
    for got, exp in zip(tr.history, ref):
        assert abs(got["lr"] - exp["lr"]) <= 1e-9 + 1e-6 * exp["lr"]
        for k in worst:
            worst[k] = max(worst[k], abs(got[k] - exp[k]) / abs(exp[k]))
        worst_g = max(worst_g, abs(got["grad_norm"] - exp["grad_norm"]) / exp["grad_norm"])
    final = load_npz("train_tiny_final.npz")
    own = dict(jm.named_parameters())
    worst_f = max(rel(own[k], v) for k, v in final.items())
    print(f"[bf16 curve vs reference] worst per-step relative loss deviation {worst}, grad norm {worst_g:.2e}, final LoRA tensors {worst_f:.2e}")
    for k, tol in BF16_CURVE_TOL.items():
        assert worst[k] < tol, (k, worst)
    assert worst_g < BF16_GNORM_TOL and worst_f < BF16_FINAL_TOL, (worst_g, worst_f)


def test_trainer_runs_on_parquet_shard(tmp_path):
    """SURVEY 8f rank 2 end to end: create_dataloader over the golden parquet shard (real on-disk schema, augmentation
    and cross-sample prompts on) -> Trainer.fit in train mode (dropouts active) on a small joint model."""
    import os
    from conftest import GOLD
    from cosyvoice_lora_finetune_framework_amd.dataset import create_dataloader
    from cosyvoice_lora_finetune_framework_amd.flow_model import build_flow_model
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.llm_model import build_llm_model
    from cosyvoice_lora_finetune_framework_amd.lora import apply_lora_to_model
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    torch.manual_seed(0)
    num = Numerics(dtype=torch.float32)
    flow = build_flow_model(None, 'cpu', numerics=num, input_size=128, vocab_size=4096, encoder_attention_heads=2,
                            encoder_linear_units=256, encoder_num_blocks=2, decoder_channels=(64, 64), decoder_attention_head_dim=64,
                            decoder_n_blocks=1, decoder_num_mid_blocks=2, decoder_num_heads=2)
    llm = build_llm_model(None, 'cpu', numerics=num, text_encoder_input_size=64, llm_input_size=128, llm_output_size=128,
                          text_token_size=512, speech_token_size=4096, attention_heads=2, linear_units=256,
                          text_encoder_blocks=2, llm_blocks=2)
    apply_lora_to_model(flow, r=4, lora_alpha=8, lora_dropout=0.05, target_modules=["to_q", "to_k", "to_v", "linear_q", "w_1"])
    apply_lora_to_model(llm, r=4, lora_alpha=8, lora_dropout=0.1, target_modules=["linear_q", "linear_k", "linear_v", "w_1", "w_2"])
    mod = JointLightningModule('joint', learning_rate=1e-3, warmup_steps=1, model=JointLLMFlowModel(llm, flow, 'joint', 2.0, 1.0),
                               numerics=num)
    loader = create_dataloader(os.path.join(GOLD, "data_shard"), batch_size=2, num_workers=0)
    assert len(loader) == 3
    # utterance 3 of the shard has no text: like the reference (llm_flow_model.py:117), a joint-mode batch without
    # text_token is an error, so train on the five utterances that have it
    from torch.utils.data import DataLoader, Subset
    from cosyvoice_lora_finetune_framework_amd.dataset import collate_fn
    loader = DataLoader(Subset(loader.dataset, [0, 1, 2, 4, 5]), batch_size=2, shuffle=True, collate_fn=collate_fn, drop_last=True)
    tr = Trainer(max_epochs=2, accumulate_grad_batches=1, default_root_dir=str(tmp_path), log_every_n_steps=1, save_checkpoints=False)
    tr.fit(mod, loader)
    assert tr.global_step == 4 and all(math.isfinite(h["loss"]) for h in tr.history)


def test_trainer_graph_path_equals_eager_path(tiny_meta):
    """Trainer.fit with the captured micro-step graph (shape seen twice -> capture, then replay from static buffers,
    text / LM lengths bucketed) reproduces the eager trainer: same losses, lr, grad-norm per step and same final LoRA
    tensors, on ragged batches with gradient accumulation (reference loop: train_joint.py:349-368)."""
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    num = Numerics(dtype=torch.float32)
    # (T, Lx, Lt): repeats, bucket-mates, and layouts that an earlier captured step covers within SHAPE_SLACK (23 -> 24, 19 -> 20)
    shapes = [(24, 5, 10), (24, 7, 10), (20, 5, 8), (23, 5, 10), (19, 5, 8), (24, 9, 10)]
    hist, finals, stats = [], [], []
    for use_graph in (False, True):
        flow = build_flow_product(tiny_meta["flow"], DEV, num)
        llm = build_llm_product(tiny_meta["llm"], DEV, num)
        jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
        module = JointLightningModule('joint', learning_rate=1e-3, min_lr=1e-5, warmup_steps=2, weight_decay=0.01, model=jm,
                                      numerics=num)
        batches = [synth_batch([T, max(4, T - 5)], text_lens=[Lx, max(2, Lx - 2)], token_lens=[Lt, max(3, Lt - 3)],
                               seed=300 + i, text_vocab=100, speech_vocab=50) for i, (T, Lx, Lt) in enumerate(shapes)]
        tr = Trainer(max_epochs=2, accumulate_grad_batches=2, gradient_clip_val=1.0, train_mode=False, log_every_n_steps=1,
                     save_checkpoints=False, use_graph=use_graph,
                     draws_fn=lambda ep, bi, b: cfm_draws(2, b["speech_feat"].shape[1], 1000 * ep + bi))
        tr.fit(module, batches)
        hist.append(tr.history)
        finals.append({k: v.detach().clone() for k, v in jm.named_parameters() if v.requires_grad})
        stats.append(tr.graph_stats)
    assert stats[0]["replays"] == 0 and stats[1]["replays"] == 12 and stats[1]["captures"] == 2 and stats[1]["eager"] == 0, stats
    assert len(hist[0]) == len(hist[1]) == 6
    for a, b in zip(*hist):
        for k in ("loss", "llm_loss", "flow_loss", "lr", "grad_norm"):
            assert abs(a[k] - b[k]) <= 1e-5 * abs(a[k]) + 1e-9, (k, a, b)
    assert max(rel(finals[1][k], finals[0][k]) for k in finals[0]) < 1e-5


def test_captured_step_with_the_optimizer_inside_equals_eager(tiny_meta):
    """accumulate_grad_batches = 1, one rank, no gradient hook: the captured micro-step ends with the optimiser step (clip, AdamW,
    bf16 shadows, gradient reset -- train_joint._StepGraph, CVFT_GRAPH_OPT), its learning rate and step count being device scalars
    set / advanced per replay.  Against the eager trainer and against the captured step WITHOUT the optimiser: per-step losses,
    learning rates, gradient norms, final LoRA tensors and the optimiser's step count, over a warm-up + cosine schedule, with a
    layout captured mid-run."""
    from cosyvoice_lora_finetune_framework_amd import train_joint as TJ
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    num = Numerics(dtype=torch.float32)
    shapes = [(24, 5, 10), (24, 7, 10), (20, 5, 8), (24, 5, 10), (20, 5, 8), (24, 9, 10)]
    runs = {}
    keep = TJ.GRAPH_OPT
    try:
        for name, use_graph, graph_opt in (("eager", False, False), ("graph", True, False), ("graph+opt", True, True)):
            TJ.GRAPH_OPT = graph_opt
            flow = build_flow_product(tiny_meta["flow"], DEV, num)
            llm = build_llm_product(tiny_meta["llm"], DEV, num)
            jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
            module = JointLightningModule('joint', learning_rate=1e-3, min_lr=1e-5, warmup_steps=3, weight_decay=0.01, model=jm, numerics=num)
            batches = [synth_batch([T, max(4, T - 5)], text_lens=[Lx, max(2, Lx - 2)], token_lens=[Lt, max(3, Lt - 3)],
                                   seed=500 + i, text_vocab=100, speech_vocab=50) for i, (T, Lx, Lt) in enumerate(shapes)]
            tr = Trainer(max_epochs=2, accumulate_grad_batches=1, gradient_clip_val=1.0, train_mode=False, log_every_n_steps=1,
                         save_checkpoints=False, use_graph=use_graph,
                         draws_fn=lambda ep, bi, b: cfm_draws(2, b["speech_feat"].shape[1], 1000 * ep + bi))
            tr.fit(module, batches)
            runs[name] = (tr.history, {k: v.detach().clone() for k, v in jm.named_parameters() if v.requires_grad},
                          tr.optimizer.step_count, float(tr.optimizer.step_dev), dict(tr.graph_stats),
                          any(g.steps_optimizer for g in tr._graphs.values()))
    finally:
        TJ.GRAPH_OPT = keep
    assert runs["graph+opt"][5] and not runs["graph"][5]                       # the optimiser really was inside / outside the graph
    assert runs["graph+opt"][4]["replays"] == 12 and runs["graph+opt"][4]["eager"] == 0, runs["graph+opt"][4]
    for name in ("graph", "graph+opt"):
        h, fin, cnt, dev_cnt = runs[name][:4]
        assert cnt == dev_cnt == runs["eager"][2] == 12, (name, cnt, dev_cnt)
        assert len(h) == len(runs["eager"][0]) == 12
        for a, b in zip(runs["eager"][0], h):
            for k in ("loss", "llm_loss", "flow_loss", "lr", "grad_norm"):
                assert abs(a[k] - b[k]) <= 1e-5 * abs(a[k]) + 1e-9, (name, k, a, b)
        assert max(rel(fin[k], runs["eager"][1][k]) for k in fin) < 1e-5, name


def test_trainer_sub_batch_chains_equal_one_chain(tiny_meta):
    """The bench's configuration in small: batches of 8 (>= 2 x SPLIT_MIN_PART, so the Flow branch runs as two concurrent
    half-batch chains on the same adapters), rank-16 adapters (matrix-core / slab LoRA-gradient path with the trainer's
    LoraGradSink), captured micro-step graph, accumulation.  Per-step losses, gradient norms and the final LoRA tensors
    equal the eager one-chain trainer's -- the reference's global-batch means (llm_flow_model.py:77-107)."""
    import copy
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    meta = copy.deepcopy(tiny_meta)
    for k in ("flow", "llm"):
        meta[k]["lora"]["r"], meta[k]["lora"]["alpha"] = 16, 32
    num = Numerics(dtype=torch.float32)
    B = 8
    hist, finals, stats = [], [], []
    saved = dict(J.SPLIT)
    try:
        for use_graph, split in ((False, {'llm': 1, 'flow': 1}), (True, {'llm': 1, 'flow': 2}), (True, {'llm': 2, 'flow': 2})):
            J.SPLIT.update(split)
            flow = build_flow_product(meta["flow"], DEV, num)
            llm = build_llm_product(meta["llm"], DEV, num)
            jm = J.JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
            module = JointLightningModule('joint', learning_rate=1e-3, min_lr=1e-5, warmup_steps=2, weight_decay=0.01, model=jm,
                                          numerics=num)
            batches = [synth_batch([24 - (i + j) % 7 for j in range(B)], text_lens=[7 - (i + j) % 3 for j in range(B)],
                                   token_lens=[12 - (2 * i + j) % 4 for j in range(B)], seed=500 + i, text_vocab=100, speech_vocab=50)
                       for i in range(4)]
            tr = Trainer(max_epochs=2, accumulate_grad_batches=2, gradient_clip_val=1.0, train_mode=False, log_every_n_steps=1,
                         save_checkpoints=False, use_graph=use_graph,
                         draws_fn=lambda ep, bi, b: cfm_draws(B, b["speech_feat"].shape[1], 1000 * ep + bi))
            tr.fit(module, batches)
            hist.append(tr.history)
            finals.append({k: v.detach().clone() for k, v in jm.named_parameters() if v.requires_grad})
            stats.append(dict(tr.graph_stats))
    finally:
        J.SPLIT.update(saved)
    assert stats[1]["replays"] == 8 and stats[1]["eager"] == 0 and stats[2]["replays"] == 8, stats
    for h in hist[1:]:
        assert len(h) == len(hist[0]) == 4
        for a, b in zip(hist[0], h):
            for k in ("loss", "llm_loss", "flow_loss", "lr", "grad_norm"):
                assert abs(a[k] - b[k]) <= 2e-5 * abs(a[k]) + 1e-9, (k, a, b)
    for f in finals[1:]:
        assert max(rel(f[k], finals[0][k]) for k in finals[0]) < 2e-5


@pytest.mark.parametrize("use_graph,big", [(0, 0), (1, 0), (1, 1)])
def test_data_parallel_trainer_equals_global_batch(tmp_path, use_graph, big):
    """SURVEY 8e with the real stack: two fresh processes (gloo, both on this GPU, CVFT_SINGLE_DEVICE=1 set before any GPU
    call) run JointLLMFlowModel + FlatAdamW + Trainer.fit with accumulation on their shards of ragged global batches.
    The all-reduced gradient norm of every optimiser step and the LoRA tensors after 4 steps equal the single-process
    global-batch run (reference semantics: global-batch means, cosyvoice/flow/flow_matching.py:192,
    label_smoothing_loss.py:91-96; no_sync on accumulation micro-steps, cosyvoice/utils/executor.py:64-65).
    big = the driver's N > 1 configuration in small: rank-16 adapters (matrix-core slab products through the LoraGradSink),
    16 utterances per global batch, so every rank's shard of 8 runs its Flow branch as two concurrent chains."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]

    def launch(rank, world, out):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CVFT_DIST_BACKEND="gloo", CVFT_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   CVFT_DPTEST_BIG=str(big))
        return subprocess.Popen([sys.executable, os.path.join(here, "dp_worker.py"), str(out), str(use_graph)], env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT)

    single = launch(0, 1, tmp_path / "single.pt")
    so, _ = single.communicate(timeout=280)
    assert single.returncode == 0, so.decode()[-2000:]
    ranks = [launch(r, 2, tmp_path / f"rank{r}.pt") for r in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in ranks]
    for p, o in zip(ranks, outs):
        assert p.returncode == 0, o.decode()[-2000:]
    ref = torch.load(tmp_path / "single.pt")
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    assert len(ref["history"]) == len(got[0]["history"]) == len(got[1]["history"]) == 4
    for a, b0, b1 in zip(ref["history"], got[0]["history"], got[1]["history"]):
        assert abs(b0["grad_norm"] - b1["grad_norm"]) <= 1e-7 * abs(b0["grad_norm"])       # identical replicas
        assert abs(a["grad_norm"] - b0["grad_norm"]) <= 2e-5 * abs(a["grad_norm"]), (a, b0)
        assert abs(a["lr"] - b0["lr"]) <= 1e-12
    for k, v in ref["params"].items():
        assert torch.equal(got[0]["params"][k], got[1]["params"][k]), k
        assert rel(got[0]["params"][k], v) < 2e-5, (k, rel(got[0]["params"][k], v))
    if use_graph:
        assert got[0]["graph_stats"]["replays"] > 0
    # no collective in front of a step: the 2-float denominator exchange of each of the 8 micro-steps ran on the prefetch thread
    # over its own (gloo, CPU tensor) group; the main thread all-reduced only the flat gradient (4 optimiser steps) and the epoch
    # metrics (2 epochs)
    for g in got:
        side = [c for c in g["allreduce_calls"] if not c[0]]
        main = [c for c in g["allreduce_calls"] if c[0]]
        assert len(side) == 8 and all(c[1] == 2 and c[2] == "cpu" and c[3] for c in side), g["allreduce_calls"]
        assert len(main) == 4 + 2 and not any(c[1] == 2 for c in main), main
