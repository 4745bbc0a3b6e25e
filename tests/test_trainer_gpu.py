"""BASELINE configs[0] (plumbing reference): 8 synthetic pairs, LoRA r=4, fp32, 2 epochs, batch 1,
accumulate 2 -- the product Trainer (flat AdamW + warmup-cosine + clip) on the HIP path must
reproduce the loss curve, learning rates, gradient norms and final LoRA tensors that the
REFERENCE's modules produced under torch.optim.AdamW / LambdaLR on the CPU
(tests/golden/train_tiny_log.json, tools/make_golden.py::gen_train)."""
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
