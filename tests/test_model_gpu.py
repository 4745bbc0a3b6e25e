"""Model-level parity on the GPU: the HIP product path (fp32 = exact-fp32 MFMA) against golden
vectors produced by the REFERENCE's own modules (tests/golden, tools/make_golden.py).
north_star tolerance: flow-matching loss and LLM CE loss within 1e-4 relative (fp32)."""
import math

import pytest
import torch

from conftest import load_json, load_npz
from helpers import build_flow_product, build_llm_product, lora_grads, rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOSS_TOL = 1e-4          # north_star: 1e-4 relative fp32
GRAD_TOL = 2e-3          # LoRA gradients (fp32 atomics + different summation order)


def _batch(g):
    return {k[3:]: v for k, v in g.items() if k.startswith("in_")}


def _numerics(variant, dtype=torch.float32):
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    return Numerics(dtype=dtype) if variant == "vendored" else Numerics.twin(dtype)


@pytest.mark.parametrize("variant", ["vendored", "twin"])
def test_flow_tiny_loss_and_grads(tiny_meta, variant):
    g = load_npz("flow_tiny.npz")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics(variant))
    draws = dict(t_raw=g["draw_t_raw"], z=g["draw_z"], cfg_rand=g["draw_cfg_rand"])
    out = m.forward_no_prompt(_batch(g), DEV, draws)
    ref = float(g[f"loss_{variant}"])
    assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL, (float(out["loss"]), ref)
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(f"grad_{variant}/")}
    assert set(grads) == set(refg)
    worst = max(rel(grads[k], refg[k]) for k in refg)
    assert worst < GRAD_TOL, worst


@pytest.mark.parametrize("variant", ["vendored", "twin"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flow_odd_padded_length_backward_matches_reference(tiny_meta, variant, dtype):
    """Reference-pinned BACKWARD with an odd padded T_max (25; ragged 25 / 18): the U-Net's ceil(T/2) down path and the
    transposed-conv up path cropped from T + 1 to T (cosyvoice/flow/decoder.py:256, 276), fixture tests/golden/flow_oddT.npz
    (tools/make_golden.py gen_oddT: the reference's own forward + autograd)."""
    g = load_npz("flow_oddT.npz")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics(variant, dtype))
    draws = dict(t_raw=g["draw_t_raw"], z=g["draw_z"], cfg_rand=g["draw_cfg_rand"])
    out = m.forward_no_prompt(_batch(g), DEV, draws)
    ref = float(g[f"loss_{variant}"])
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(f"grad_{variant}/")}
    assert set(grads) == set(refg)
    if dtype == torch.float32:
        assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL, (float(out["loss"]), ref)
        assert max(rel(grads[k], refg[k]) for k in refg) < GRAD_TOL
    else:
        assert abs(float(out["loss"]) - ref) / ref < 3e-2
        num = math.sqrt(sum(float(((grads[k].double().cpu() - refg[k].double()) ** 2).sum()) for k in refg))
        den = math.sqrt(sum(float((refg[k].double() ** 2).sum()) for k in refg))
        assert num / den < 0.1


def test_flow_tiny_bf16_close(tiny_meta):
    g = load_npz("flow_tiny.npz")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics("vendored", torch.bfloat16))
    draws = dict(t_raw=g["draw_t_raw"], z=g["draw_z"], cfg_rand=g["draw_cfg_rand"])
    out = m.forward_no_prompt(_batch(g), DEV, draws)
    ref = float(g["loss_vendored"])
    assert abs(float(out["loss"]) - ref) / ref < 3e-2
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("grad_vendored/")}
    num = math.sqrt(sum(float(((grads[k].double().cpu() - refg[k].double()) ** 2).sum()) for k in refg))
    den = math.sqrt(sum(float((refg[k].double() ** 2).sum()) for k in refg))
    assert num / den < 0.1


def test_llm_tiny_loss_and_grads(tiny_meta):
    g = load_npz("llm_tiny.npz")
    m = build_llm_product(tiny_meta["llm"], DEV, _numerics("vendored"))
    out = m.forward_no_prompt(_batch(g), DEV)
    ref = float(g["loss"])
    assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL, (float(out["loss"]), ref)
    assert abs(float(out["acc"]) - float(g["acc"])) < 1e-6
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("grad/")}
    assert set(grads) == set(refg)
    worst = max(rel(grads[k], refg[k]) for k in refg)
    assert worst < GRAD_TOL, worst


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llm_default_lora_targets_train_linear_pos(dtype):
    """apply_lora_to_model(model) with the DEFAULT target list (reference lora.py:155-166) wraps linear_pos; the rel-pos
    attention backward then returns the gradient w.r.t. the projected positional encoding (cvft_attn_relpos_bwd dp).
    fp32: loss 1e-4 and all LoRA gradients (linear_pos included) vs the reference run; bf16: aggregate closeness."""
    from conftest import load_json
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    g = load_npz("llm_tiny_pos.npz")
    lm = load_json("tiny_pos_meta.json")["llm"]
    num = Numerics(dtype=dtype)
    m = build_llm_product(lm, DEV, num)
    assert type(m.llm.encoders[0].self_attn.linear_pos).__name__ == "LoRALinear"
    assert [k for k, _ in m.state_dict().items()] == [k for k, _ in lm["spec"]]      # same key set / order as the reference
    out = m.forward_no_prompt(_batch(g), DEV)
    ref = float(g["loss"])
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("grad/")}
    assert set(grads) == set(refg)
    if dtype == torch.float32:
        assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL, (float(out["loss"]), ref)
        worst = max(rel(grads[k], refg[k]) for k in refg)
        assert worst < GRAD_TOL, worst
        assert max(rel(grads[k], refg[k]) for k in refg if "linear_pos" in k) < GRAD_TOL
    else:
        assert abs(float(out["loss"]) - ref) / ref < 2e-2
        num_ = sum(float((grads[k].double().cpu() - refg[k].double()).norm() ** 2) for k in refg) ** 0.5
        den_ = sum(float(refg[k].double().norm() ** 2) for k in refg) ** 0.5
        assert num_ / den_ < 0.1, num_ / den_


def test_joint_model_contract(tiny_meta):
    """JointLLMFlowModel.forward dict contract + loss weighting (llm_flow_model.py:77-107)."""
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    gf, gl = load_npz("flow_tiny.npz"), load_npz("llm_tiny.npz")
    flow = build_flow_product(tiny_meta["flow"], DEV, _numerics("vendored"))
    llm = build_llm_product(tiny_meta["llm"], DEV, _numerics("vendored"))
    jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
    draws = dict(t_raw=gf["draw_t_raw"], z=gf["draw_z"], cfg_rand=gf["draw_cfg_rand"])
    out = jm(_batch(gf), torch.device(DEV), draws)
    assert set(out) == {"loss", "llm_loss", "flow_loss", "llm_acc"}
    ref = 2.0 * float(gl["loss"]) + float(gf["loss_vendored"])
    assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL
    for mode, keys in (("llm_only", {"loss", "llm_loss", "llm_acc"}), ("flow_only", {"loss", "flow_loss"})):
        jm.training_mode = mode
        assert set(jm(_batch(gf), torch.device(DEV), draws)) == keys


def test_merged_export_roundtrip(tiny_meta):
    """lora.get_merged_state_dict: original key set, strict-loadable into an un-wrapped model, same loss
    (reference lora.py:284-323; consumer inference_joint.py:113-127)."""
    from cosyvoice_lora_finetune_framework_amd.lora import get_merged_state_dict
    g, gm = load_npz("flow_tiny.npz"), load_npz("flow_tiny_merged.npz")
    fm = tiny_meta["flow"]
    m = build_flow_product(fm, DEV, _numerics("vendored"))
    draws = dict(t_raw=g["draw_t_raw"], z=g["draw_z"], cfg_rand=g["draw_cfg_rand"])
    with torch.no_grad():
        l0 = float(m.forward_no_prompt(_batch(g), DEV, draws)["loss"])
    merged = get_merged_state_dict(m)
    assert sorted(merged.keys()) == fm["base_keys"]
    for k, v in gm.items():
        assert rel(merged[k], v) < 1e-6, k
    base = build_flow_product(fm, "cpu", _numerics("vendored"), lora=False)
    base.load_state_dict({k: v.cpu() for k, v in merged.items()}, strict=True)
    base = base.to(DEV)
    with torch.no_grad():
        l1 = float(base.forward_no_prompt(_batch(g), DEV, draws)["loss"])
    assert abs(l1 - l0) / l0 < 2e-5
    assert abs(l1 - fm["merged_loss"]) / fm["merged_loss"] < LOSS_TOL


@pytest.mark.parametrize("case", ["uniform_T500_B2", "ragged_T500_B2"])
def test_full_size_flow_matches_reference(case):
    """CosyVoice-300M flow dims, BASELINE shape (T=500, r=16): fp32 loss within 1e-4 of the reference CPU
    run; LoRA grad norm within 1e-3; bf16 loss within 2e-2."""
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    fs = load_json("full_scalars.json")
    c = fs[f"flow/{case}"]
    meta = dict(lora=dict(r=fs["flow_lora"]["r"], alpha=fs["flow_lora"]["alpha"], targets=fs["flow_lora"]["targets"]),
                weight_seed=fs["flow_lora"]["weight_seed"])
    m = build_flow_product(meta, DEV, _numerics("vendored"))
    batch = synth_batch(c["feat_lens"], text_lens=c["text_lens"], seed=c["batch_seed"])
    draws = cfm_draws(len(c["feat_lens"]), max(c["feat_lens"]), c["draw_seed"])
    out = m.forward_no_prompt(batch, DEV, draws)
    assert abs(float(out["loss"]) - c["loss"]) / c["loss"] < LOSS_TOL, (float(out["loss"]), c["loss"])
    out["loss"].backward()
    grads = lora_grads(m)
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    assert abs(tot - c["grads"]["total_norm"]) / c["grads"]["total_norm"] < 1e-3
    for k, v in c["grads"]["picks"].items():
        assert abs(float(grads[k].double().norm()) - v) / v < 2e-3, k
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    m.zero_grad(set_to_none=True)
    m.numerics = Numerics(dtype=torch.bfloat16)
    with torch.no_grad():
        lb = float(m.forward_no_prompt(batch, DEV, draws)["loss"])
    assert abs(lb - c["loss"]) / c["loss"] < 2e-2, lb


@pytest.mark.parametrize("case", ["uniform_T500_B2", "ragged_T500_B2"])
def test_full_size_llm_matches_reference(case):
    from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch
    fs = load_json("full_scalars.json")
    c = fs[f"llm/{case}"]
    meta = dict(lora=dict(r=fs["llm_lora"]["r"], alpha=fs["llm_lora"]["alpha"], targets=fs["llm_lora"]["targets"]),
                weight_seed=fs["llm_lora"]["weight_seed"])
    m = build_llm_product(meta, DEV, _numerics("vendored"), full=True)
    batch = synth_batch(c["feat_lens"], text_lens=c["text_lens"], seed=c["batch_seed"])
    out = m.forward_no_prompt(batch, DEV)
    assert abs(float(out["loss"]) - c["loss"]) / c["loss"] < LOSS_TOL, (float(out["loss"]), c["loss"])
    assert abs(float(out["acc"]) - c["acc"]) < 1e-6
    out["loss"].backward()
    grads = lora_grads(m)
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    assert abs(tot - c["grads"]["total_norm"]) / c["grads"]["total_norm"] < 1e-3
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    m.zero_grad(set_to_none=True)
    m.numerics = Numerics(dtype=torch.bfloat16)
    with torch.no_grad():
        lb = float(m.forward_no_prompt(batch, DEV)["loss"])
    assert abs(lb - c["loss"]) / c["loss"] < 2e-2, lb


# Stated tolerances of the reduced-precision BACKWARD at full size (relative L2 against the reference's fp32 CPU gradients) --
# each <= 3x the value measured on MI355X in round 4 (printed by _check_backward):
#   bf16: loss 3.2e-5 .. 3.7e-4, total gradient norm 3.5e-4 .. 5.7e-3, worst picked adapter tensor 2.3e-2 .. 3.8e-2 on the flow
#         (T = 500 / r = 16 and T = 1000 / r = 64) and 6.7e-2 .. 7.3e-2 on the LLM;  fp8 (LLM, T = 1000): loss 1.06e-4, norm 1.3e-3;
#         fp32: 1e-6 / 5e-6 / 8e-4.
# Round 3 had 2.1e-1 on the to_q / to_k adapters of the estimator's MID blocks at T = 1000 / r = 64 (tensors holding 3e-4 .. 5e-4 of
# the gradient norm): with ~500 keys per query the softmax is nearly flat, dP is nearly constant over the keys, and the score
# gradient dS = P (dP - delta) inherited the error of delta = rowsum(dO . O) taken from the bf16-ROUNDED forward output as a
# common-mode error of the whole row (tools/delta_error_model.py: 3 % .. 10 % on dQ / dK in that regime).  The forward now also
# writes O's rounding residual and the backward forms delta from O + residual (include/cvft.h `o_lo`, csrc/attn_common.h): the same
# tensors are at 3.8e-2 (CVFT_ATTN_OLO=0 restores the old form: 1.1e-1 on this build), and one per-tensor bound serves tensors of
# every share of the norm.
BF16_LOSS_TOL, BF16_GRAD_NORM_TOL = 1.1e-3, 1.5e-2
BF16_GRAD_TENSOR_TOL, BF16_SMALL_TENSOR_TOL, SMALL_SHARE = 1.1e-1, 1.1e-1, 2e-3
FP8_LOSS_TOL, FP8_GRAD_NORM_TOL = 5e-4, 4e-3


def _full_case(branch, tag):
    fm = load_json("full_grads_meta.json")
    c = fm[f"{branch}/{tag}"]
    ref = {k.split("/", 2)[2]: v for k, v in load_npz("full_grads.npz").items() if k.startswith(f"{branch}/{tag}/")}
    return c, ref


def _trainer_style_backward(m, fwd, c, ref):
    """The backward as train_joint.Trainer / bench.py run it: LoRA masters and gradients in FlatAdamW's flat fp32 buffers
    (bf16 shadows, stacked q|k|v operands), backward inside a LoraGradSink (matrix-core slab products + ONE reduce
    launch); eval() like the reference fixtures (dropout off)."""
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    params = [p for p in m.parameters() if p.requires_grad]
    opt = FlatAdamW(params, lr=1e-4)
    opt.zero_grad()
    out = fwd()
    with HF.LoraGradSink():
        out["loss"].backward()
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in params)
    _check_backward(lora_grads(m), c, ref, float(out["loss"]), "bf16")


def _check_backward(grads, c, ref, loss, mode):
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    rn = abs(tot - c["grads"]["total_norm"]) / c["grads"]["total_norm"]
    rl = abs(loss - c["loss"]) / c["loss"]
    worst = max(rel(grads[k], v) for k, v in ref.items())
    # tensors that carry a visible share of the gradient norm / the tiny ones (see the note above)
    share = {k: float(v.double().norm()) / c["grads"]["total_norm"] for k, v in ref.items()}
    worst_big = max([rel(grads[k], v) for k, v in ref.items() if share[k] >= SMALL_SHARE] or [0.0])
    worst_small = max([rel(grads[k], v) for k, v in ref.items() if share[k] < SMALL_SHARE] or [0.0])
    print(f"[{mode}] loss rel {rl:.2e}  grad-norm rel {rn:.2e}  worst picked tensor rel-L2 {worst:.2e} "
          f"(norm share >= {SMALL_SHARE}: {worst_big:.2e}, smaller: {worst_small:.2e})")
    assert set(ref) <= set(grads)
    if mode == "fp32":
        assert rl < LOSS_TOL and rn < 1e-3 and worst < GRAD_TOL, (rl, rn, worst)
    elif mode == "bf16":
        assert rl < BF16_LOSS_TOL and rn < BF16_GRAD_NORM_TOL, (rl, rn)
        assert worst_big < BF16_GRAD_TENSOR_TOL and worst_small < BF16_SMALL_TENSOR_TOL, (worst_big, worst_small)
    else:
        assert rl < FP8_LOSS_TOL and rn < FP8_GRAD_NORM_TOL, (rl, rn, worst)


@pytest.mark.parametrize("tag", ["t500_r16", "t1000_r64"])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16_trainer"])
def test_full_size_flow_backward_matches_reference(tag, mode):
    """CosyVoice-300M flow, ragged B=2, BASELINE configs[1..3] shape (T=500, r=16) and configs[4] shape (T=1000, r=64):
    loss, total LoRA gradient norm and whole gradient tensors of six adapters (first / middle / last, A and B) against
    the reference's fp32 CPU run -- fp32 path to 1e-4 / 2e-3, and the bf16 path (the arithmetic bench.py times:
    stacked q|k|v, fused feed-forward, 32x32 MFMA attention backward) to the tolerances stated above."""
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    c, ref = _full_case("flow", tag)
    meta = dict(lora=dict(r=c["r"], alpha=c["alpha"], targets=load_json("full_scalars.json")["flow_lora"]["targets"]),
                weight_seed=c["weight_seed"])
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    m = build_flow_product(meta, DEV, _numerics("vendored", dt))
    batch = synth_batch(c["feat_lens"], text_lens=c["text_lens"], seed=c["batch_seed"])
    draws = cfm_draws(len(c["feat_lens"]), max(c["feat_lens"]), c["draw_seed"])
    if mode == "bf16_trainer":
        _trainer_style_backward(m, lambda: m.forward_no_prompt(batch, DEV, draws), c, ref)
        return
    out = m.forward_no_prompt(batch, DEV, draws)
    out["loss"].backward()
    _check_backward(lora_grads(m), c, ref, float(out["loss"]), mode)


@pytest.mark.parametrize("tag", ["t500_r16", "t1000_r64"])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp8", "bf16_trainer"])
def test_full_size_llm_backward_matches_reference(tag, mode):
    """Same for the LLM branch (L = 333 / 623); "fp8" = BASELINE configs[4] arithmetic (frozen-W GEMMs of the LLM-sized
    linears in OCP e4m3 with per-token / per-channel scales, everything else bf16)."""
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch
    if mode == "fp8" and tag != "t1000_r64":
        pytest.skip("fp8 is the configs[4] arithmetic")
    c, ref = _full_case("llm", tag)
    meta = dict(lora=dict(r=c["r"], alpha=c["alpha"], targets=load_json("full_scalars.json")["llm_lora"]["targets"]),
                weight_seed=c["weight_seed"])
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    m = build_llm_product(meta, DEV, _numerics("vendored", dt), full=True)
    batch = synth_batch(c["feat_lens"], text_lens=c["text_lens"], seed=c["batch_seed"])
    if mode == "bf16_trainer":
        _trainer_style_backward(m, lambda: m.forward_no_prompt(batch, DEV), c, ref)
        return
    HF.FP8_ON = mode == "fp8"
    try:
        out = m.forward_no_prompt(batch, DEV)
        out["loss"].backward()
    finally:
        HF.FP8_ON = False
    assert abs(float(out["acc"]) - c["acc"]) < (1e-6 if mode == "fp32" else 2e-2)
    _check_backward(lora_grads(m), c, ref, float(out["loss"]), mode)


@pytest.mark.parametrize("branch,B", [("flow", 8), ("llm", 16)])
def test_bench_batch_equals_accumulated_small_batches(branch, B):
    """The bench's batch sizes on the bench's code path (CosyVoice-300M dims, bf16, FlatAdamW flat buffers, LoraGradSink,
    sub-batch chains: Flow 2 x 4 utterances, LLM-only 2 x 8) against the SAME utterances run two at a time -- the batch size
    the reference-pinned full-size tests use -- with gradient accumulation.  Uniform lengths, so every utterance's
    contribution is independent of its batch mates (GroupNorm statistics are per utterance): loss and accumulated LoRA
    gradients must agree to bf16 rounding.  Catches anything indexed by batch that B = 2 cannot (block maps over
    batch x head, row-block plans, slab bookkeeping)."""
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    fs = load_json("full_scalars.json")
    L = fs[f"{branch}_lora"]
    meta = dict(lora=dict(r=L["r"], alpha=L["alpha"], targets=L["targets"]), weight_seed=L["weight_seed"])
    num = _numerics("vendored", torch.bfloat16)
    if branch == "flow":
        flow, llm, mode = build_flow_product(meta, DEV, num), torch.nn.Identity(), "flow_only"
    else:
        flow = build_flow_product(dict(meta, lora=dict(meta["lora"], targets=["to_q"])), DEV, num)     # (unused by llm_only)
        flow.requires_grad_(False)
        llm, mode = build_llm_product(meta, DEV, num, full=True), "llm_only"
    jm = J.JointLLMFlowModel(llm, flow, mode, 2.0, 1.0).to(DEV).eval()
    T = 500
    batch = synth_batch([T] * B, seed=77)
    draws = cfm_draws(B, T, 9)
    params = [p for p in jm.parameters() if p.requires_grad]
    opt = FlatAdamW(params, lr=1e-4)

    def run(groups):
        opt.zero_grad()
        total = 0.0
        for sl in groups:
            sub = {k: v[sl] for k, v in batch.items()}
            d = {k: v[sl] for k, v in draws.items()}
            out = jm(sub, DEV, d if branch == "flow" else None)
            w = (sl.stop - sl.start) / B
            with HF.LoraGradSink():
                (out["loss"] * w).backward()
            total += float(out["loss"]) * w
        torch.cuda.synchronize()
        return total, opt.flat_g.clone()
    min_part_fo, J.SPLIT_MIN_PART_FLOW_ONLY = J.SPLIT_MIN_PART_FLOW_ONLY, 4      # (flow_only splits from 2 x 8 by default)
    try:
        big_loss, big = run([slice(0, B)])
    finally:
        J.SPLIT_MIN_PART_FLOW_ONLY = min_part_fo
    small_loss, small = run([slice(i, i + 2) for i in range(0, B, 2)])
    assert abs(big_loss - small_loss) / abs(small_loss) < 2e-3, (big_loss, small_loss)
    assert float(small.norm()) > 0
    assert rel(big, small) < 2e-2, rel(big, small)
    off, worst = 0, (0.0, None)
    for n_, p in [(n_, p) for n_, p in jm.named_parameters() if p.requires_grad]:
        k = p.numel()
        if float(small[off:off + k].norm()) > 1e-3 * float(small.norm()):
            worst = max(worst, (rel(big[off:off + k], small[off:off + k]), n_))
        off += k
    assert worst[0] < 0.15, worst


def test_conformer_convolution_module_matches_reference():
    """ConvolutionModule (pointwise -> GLU -> depthwise k=15 -> LayerNorm -> SiLU -> pointwise) vs the
    vendored cosyvoice/transformer/convolution.py output (ops.npz), both dtypes."""
    from oracle.detweights import det_state_dict
    from cosyvoice_lora_finetune_framework_amd.modules import ConvolutionModule
    g, m = load_npz("ops.npz"), load_json("ops_meta.json")
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 3e-2)):
        cm = ConvolutionModule(32, 15, "silu", "layer_norm", causal=False)
        cm.load_state_dict(det_state_dict([(k, tuple(s)) for k, s in m["convmod_spec"]], m["convmod_seed"]), strict=True)
        cm = cm.to(DEV)
        x = g["convmod_x"]
        B, T, Cc = x.shape
        length = g["convmod_mask"].reshape(B, T).sum(1).to(torch.int32).to(DEV)
        y = cm.forward_cl(x.reshape(B * T, Cc).to(DEV, dtype), B, T, length)
        assert rel(y.reshape(B, T, Cc), g["convmod_y"]) < tol


def test_lora_conv1d_matches_reference():
    """LoRAConv1d (1x1 conv == Linear over channels) forward vs reference lora.py:121-131."""
    from oracle.detweights import det_state_dict
    from cosyvoice_lora_finetune_framework_amd.lora import LoRAConv1d
    g, m = load_npz("ops.npz"), load_json("ops_meta.json")
    lc = LoRAConv1d(torch.nn.Conv1d(12, 20, 1), r=4, lora_alpha=8, lora_dropout=0.0)
    lc.load_state_dict(det_state_dict([(k, tuple(s)) for k, s in m["loraconv_spec"]], m["loraconv_seed"]), strict=True)
    lc = lc.to(DEV).eval()
    y = lc(g["loraconv_x"].to(DEV))
    assert rel(y, g["loraconv_y"]) < 2e-5


def test_lora_linear_module_any_shape_and_dropout():
    """LoRALinear.forward is a drop-in for arbitrary leading dims (lora.py:64-76); in train() mode the LoRA-path
    dropout changes the output, in eval() it is deterministic."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    lin = torch.nn.Linear(24, 40)
    ll = LoRALinear(lin, r=4, lora_alpha=8, lora_dropout=0.5).to(DEV)
    x = torch.randn(3, 5, 24, device=DEV)
    ll.eval()
    y0, y1 = ll(x), ll(x)
    assert y0.shape == (3, 5, 40) and torch.equal(y0, y1)
    ref = torch.nn.functional.linear(x, lin.weight.to(DEV), lin.bias.to(DEV)) + 2.0 * (x @ ll.lora_A.t()) @ ll.lora_B.t()
    assert rel(y0, ref) < 2e-5
    ll.train()
    assert not torch.equal(ll(x), y0)


def test_checkpoint_roundtrip_lightning_layout(tiny_meta, tmp_path):
    """Trainer checkpoints use the Lightning key layout (model.llm.* / model.flow.*, merge_joint_weights.py:95-104)
    and restore parameters + optimiser state (resume, train_joint.py:364-368)."""
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    num = Numerics(dtype=torch.float32)

    def make():
        flow = build_flow_product(tiny_meta["flow"], DEV, num)
        llm = build_llm_product(tiny_meta["llm"], DEV, num)
        return JointLightningModule('joint', learning_rate=1e-3, warmup_steps=1, model=JointLLMFlowModel(llm, flow, 'joint', 2.0, 1.0),
                                    numerics=num)
    batches = [synth_batch([20 + i], text_lens=[5], token_lens=[9 + i], seed=i, text_vocab=100, speech_vocab=50) for i in range(2)]
    mod = make()
    tr = Trainer(max_epochs=1, accumulate_grad_batches=1, default_root_dir=str(tmp_path), log_every_n_steps=1, train_mode=False)
    tr.fit(mod, batches)
    ck = torch.load(str(tmp_path / "joint_joint_last.ckpt"), map_location="cpu")
    assert all(k.startswith("model.llm.") or k.startswith("model.flow.") for k in ck["state_dict"])
    assert ck["global_step"] == 2
    mod2 = make()
    tr2 = Trainer(max_epochs=1, default_root_dir=str(tmp_path), save_checkpoints=False, train_mode=False)
    mod2.setup()
    opt2 = mod2.configure_optimizers()
    tr2.load_checkpoint(mod2, opt2, str(tmp_path / "joint_joint_last.ckpt"))
    a = dict(mod.model.named_parameters())
    for k, v in mod2.model.named_parameters():
        assert torch.equal(v.detach().cpu(), a[k].detach().cpu()), k
    assert torch.equal(opt2.m.cpu(), tr.optimizer.m.cpu()) and opt2.step_count == 2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_cfm_sampler_matches_reference(tiny_meta, dtype, tol):
    """the CFM Euler sampler (flow_model.py:74-135) on the HIP estimator vs the reference's output."""
    g = load_npz("sampler_tiny.npz")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics("vendored", dtype))
    out, cache = m.decoder(g["mu"].to(DEV), g["mask"].to(DEV), n_timesteps=5, temperature=1.0, spks=g["spks"].to(DEV),
                           cond=g["cond"].to(DEV), prompt_len=10, noise=g["z"], num=_numerics("vendored", dtype))
    assert out.dtype == torch.float32 and tuple(out.shape) == tuple(g["out"].shape)
    assert rel(out, g["out"]) < tol, rel(out, g["out"])
    assert rel(cache, g["cache"]) < 1e-6


def test_sub_batch_chains_reproduce_global_batch_means(tiny_meta):
    """JointLLMFlowModel can run LLM / Flow x sub-batches as independent chains on separate streams; the recombined
    losses (weighted by each part's share of frames / target tokens) and the LoRA gradients equal the one-chain step."""
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    num = Numerics(dtype=torch.float32)
    flow = build_flow_product(tiny_meta["flow"], DEV, num)
    llm = build_llm_product(tiny_meta["llm"], DEV, num)
    jm = J.JointLLMFlowModel(llm, flow, 'joint', 2.0, 1.0).to(DEV).eval()
    batch = synth_batch([24, 17, 21], text_lens=[7, 5, 6], token_lens=[13, 9, 11], seed=11, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(3, 24, seed=5)
    outs = []
    for split in ({'llm': 1, 'flow': 1}, {'llm': 2, 'flow': 2}, {'llm': 1, 'flow': 3}):
        J.SPLIT.update(split)
        min_part, J.SPLIT_MIN_PART = J.SPLIT_MIN_PART, 1
        try:
            for p in jm.parameters():
                p.grad = None
            out = jm(batch, DEV, draws)
            out['loss'].backward()
            outs.append(([float(out[k]) for k in ('loss', 'llm_loss', 'flow_loss', 'llm_acc')],
                         {n: p.grad.clone() for n, p in jm.named_parameters() if p.grad is not None}))
        finally:
            J.SPLIT.update({'llm': 1, 'flow': 2})
            J.SPLIT_MIN_PART = min_part
    for o in outs[1:]:
        for a, b in zip(outs[0][0], o[0]):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (outs[0][0], o[0])
        assert set(outs[0][1]) == set(o[1])
        assert max(rel(o[1][k], outs[0][1][k]) for k in outs[0][1]) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sub_batch_chains_with_gradient_sink(tiny_meta, dtype):
    """The trainer's configuration of the same thing: rank-16 adapters, pre-allocated fp32 .grad buffers and an active
    LoraGradSink (slab workspaces + one reduce) while the sub-batch chains of a branch run concurrently on the SAME
    adapters -- every chain must own its slabs.  Gradients equal the one-chain step's."""
    import copy
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    meta = copy.deepcopy(tiny_meta)
    for k in ("flow", "llm"):
        meta[k]["lora"]["r"], meta[k]["lora"]["alpha"] = 16, 32
    num = Numerics(dtype=dtype)
    flow = build_flow_product(meta["flow"], DEV, num)
    llm = build_llm_product(meta["llm"], DEV, num)
    jm = J.JointLLMFlowModel(llm, flow, 'joint', 2.0, 1.0).to(DEV).eval()
    batch = synth_batch([24, 17, 21, 19], text_lens=[7, 5, 6, 4], token_lens=[13, 9, 11, 10], seed=11, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(4, 24, seed=5)
    params = [(n, p) for n, p in jm.named_parameters() if p.requires_grad]
    outs = []
    for split in ({'llm': 1, 'flow': 1}, {'llm': 2, 'flow': 2}, {'llm': 1, 'flow': 4}):
        J.SPLIT.update(split)
        min_part, J.SPLIT_MIN_PART = J.SPLIT_MIN_PART, 1
        try:
            # twice: the first backward of a split meets more products per adapter than the slab buffer was sized for
            # (separate buffers, reduce tasks in successive launches), the second finds one contiguous range (merged tasks)
            for rep in range(2):
                for _, p in params:
                    p.grad = torch.zeros_like(p, dtype=torch.float32)
                out = jm(batch, DEV, draws)
                with HF.LoraGradSink():
                    out['loss'].backward()
                torch.cuda.synchronize()
                outs.append({n: p.grad.clone() for n, p in params})
        finally:
            J.SPLIT.update({'llm': 1, 'flow': 2})
            J.SPLIT_MIN_PART = min_part
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    for o in outs[1:]:
        worst = max((rel(o[k], outs[0][k]), k) for k in outs[0] if float(outs[0][k].norm()) > 0)
        assert worst[0] < tol, worst


def test_shape_bucketed_batch_equals_exact_batch(tiny_meta):
    """train_joint.Trainer._fit_layout: a batch padded up to the layout of another captured step (T, Lt, text, LM length all
    larger, `_true_dims` = the exact maxima) gives the exact batch's losses and LoRA gradients -- the length regulator's
    interpolation and every GroupNorm work from the device scalars, everything else is masked by the length vectors."""
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import Trainer
    num = Numerics(dtype=torch.float32)
    flow = build_flow_product(tiny_meta["flow"], DEV, num)
    llm = build_llm_product(tiny_meta["llm"], DEV, num)
    jm = J.JointLLMFlowModel(llm, flow, 'joint', 2.0, 1.0).to(DEV).eval()
    batch = synth_batch([23, 17, 21], text_lens=[7, 5, 6], token_lens=[13, 9, 11], seed=11, text_vocab=100, speech_vocab=50)
    T = batch['speech_feat'].shape[1]
    draws = cfm_draws(3, T, seed=5)
    tr = Trainer(use_graph=False)
    outs = []
    for layout in (None, (T + 2, batch['speech_token'].shape[1] + 3, 32, 48, 3)):      # within SHAPE_SLACK of the batch
        if layout is None:                      # the batch as the loader made it: exact shapes, no `_true_dims`
            fitted, dims = batch, (T, 0, 0, 0, 3)
        else:
            tr._layouts.append(layout)
            fitted, dims = tr._fit_layout(batch)
            assert dims == layout and fitted['speech_feat'].shape[1] == T + 2 and int(fitted['_true_dims'][1]) == T
        prepared = jm.prepare_batch(fitted, DEV, 16, dims[3])
        d = dict(draws)
        if d['z'].shape[-1] < dims[0]:
            d['z'] = torch.nn.functional.pad(d['z'], (0, dims[0] - d['z'].shape[-1]))
        for p in jm.parameters():
            p.grad = None
        out = jm(prepared, DEV, d)
        out['loss'].backward()
        outs.append(([float(out[k]) for k in ('loss', 'llm_loss', 'flow_loss', 'llm_acc')],
                     {n: p.grad.clone() for n, p in jm.named_parameters() if p.grad is not None}))
    for a, b in zip(outs[0][0], outs[1][0]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (outs[0][0], outs[1][0])
    assert set(outs[0][1]) == set(outs[1][1])
    assert max(rel(outs[1][1][k], outs[0][1][k]) for k in outs[0][1]) < 2e-5


def test_flow_prompt_path_loss_and_grads(tiny_meta):
    """MaskedDiffWithXvec.forward with prompts and the anti-leakage strategies (flow_model.py:248-400,
    137-204) -- the product's forward() under the reference's `random` seed reproduces the reference's per-utterance
    decisions (dropout / dynamic length / cross-sample / silence band / text blinding), loss and LoRA gradients."""
    import random
    from cosyvoice_lora_finetune_framework_amd import flow_model as FM
    g, meta = load_npz("flow_prompt_tiny.npz"), load_json("flow_prompt_meta.json")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics("vendored"))
    draws = dict(t_raw=g["draw_t_raw"], z=g["draw_z"], cfg_rand=g["draw_cfg_rand"])
    old = (FM.ANTI_LEAKAGE_CONFIG, FM.NO_PROMPT_TRAINING_CONFIG)
    FM.ANTI_LEAKAGE_CONFIG, FM.NO_PROMPT_TRAINING_CONFIG = meta["anti_leakage"], {"enabled": False}
    seen = {}
    orig = m.decoder.compute_loss_cl

    def spy(*a, **kw):
        seen.update(prompt_lens=list(kw["prompt_lens"]), cond=kw["cond"].detach().clone(), mu=a[1].detach().clone())
        return orig(*a, **kw)
    m.decoder.compute_loss_cl = spy
    try:
        random.seed(meta["random_seed"])
        out = m.forward_with_prompt(_batch(g), DEV, draws)
    finally:
        FM.ANTI_LEAKAGE_CONFIG, FM.NO_PROMPT_TRAINING_CONFIG = old
    B, _, T = g["cond"].shape
    assert seen["prompt_lens"] == g["prompt_lens"].tolist()
    assert rel(seen["cond"].reshape(B, T, 80).transpose(1, 2), g["cond"]) < 1e-6
    assert rel(seen["mu"].reshape(B, T, 80).transpose(1, 2), g["mu"]) < 1e-4
    ref = float(g["loss"])
    assert abs(float(out["loss"]) - ref) / ref < LOSS_TOL, (float(out["loss"]), ref)
    out["loss"].backward()
    grads = lora_grads(m)
    refg = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith("grad/")}
    assert set(grads) == set(refg)
    worst = max(rel(grads[k], refg[k]) for k in refg)
    assert worst < GRAD_TOL, worst
    assert m.decoder.estimator.prompt_isolation_len == 0          # reset after the step (flow_model.py:176-177)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 6e-2)])
def test_flow_inference_entries_match_reference(tiny_meta, dtype, tol):
    """MaskedDiffWithXvec.inference (flow_model.py:474-551) and inference_like_training (553-638) on the
    HIP path against the reference's outputs (initial noise pinned)."""
    g = load_npz("flow_inference_tiny.npz")
    m = build_flow_product(tiny_meta["flow"], DEV, _numerics("vendored", dtype))
    one = lambda n: torch.tensor([n])
    mel, cache = m.inference(g["token"], one(46), g["prompt_token"], one(12), g["prompt_feat"], one(20), g["embedding"], noise=g["z"])
    assert mel.dtype == torch.float32 and tuple(mel.shape) == tuple(g["inf_mel"].shape)
    assert rel(mel, g["inf_mel"]) < tol, rel(mel, g["inf_mel"])
    assert rel(cache, g["inf_cache"]) < (1e-5 if dtype == torch.float32 else 2e-2)
    a = m.inference_like_training(g["token"], one(46), 52, g["embedding"], prompt_feat=g["prompt_feat"], prompt_len=9, noise=g["z2"])
    b = m.inference_like_training(g["token"], one(46), one(52), g["embedding"], n_timesteps=4, noise=g["z2"])
    assert rel(a, g["ilt_mel"]) < tol and rel(b, g["ilt_mel_noprompt"]) < tol, (rel(a, g["ilt_mel"]), rel(b, g["ilt_mel_noprompt"]))


def test_flow_no_prompt_mixed_mode(tiny_meta):
    """flow_model.py:437-455: NO_PROMPT 'mixed' mode = per-utterance short own-mel prompts drawn from `random`, routed through
    the prompt path's loss mask; checked against the oracle on the plan re-drawn under the same seed."""
    import random
    from oracle import ref_math as R
    from oracle.detweights import det_state_dict
    from cosyvoice_lora_finetune_framework_amd import flow_model as FM
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    fm = tiny_meta["flow"]
    m = build_flow_product(fm, DEV, _numerics("vendored"))
    batch = synth_batch([40, 33, 28], text_lens=[7, 5, 6], token_lens=[22, 18, 15], seed=4, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(3, 40, seed=9)
    old = FM.NO_PROMPT_TRAINING_CONFIG
    FM.NO_PROMPT_TRAINING_CONFIG = {"enabled": True, "mode": "mixed", "no_prompt_ratio": 0.4}
    seen = {}
    orig_fw, orig_draws = m.forward_with_prompt, m.decoder.make_draws
    m.forward_with_prompt = lambda b, d, dr=None, plan=None: (seen.update(plan=plan), orig_fw(b, d, draws, plan))[1]
    try:
        random.seed(11)
        out = m(batch, DEV)
    finally:
        FM.NO_PROMPT_TRAINING_CONFIG = old
    random.seed(11)
    want = [0 if random.random() < 0.4 else random.randint(1, max(2, int(0.1 * j))) for j in [40, 33, 28]]
    assert [p["total"] for p in seen["plan"]] == want and any(want) and not all(want)
    sd = det_state_dict([(k, tuple(s)) for k, s in fm["spec"]], fm["weight_seed"])
    cfg = R.OracleConfig(flow_lora_scale=fm["lora"]["alpha"] / fm["lora"]["r"])
    ref = R.flow_forward_prompt(sd, batch, draws, cfg, seen["plan"], FM.ANTI_LEAKAGE_CONFIG["boundary_frames"],
                                FM.ANTI_LEAKAGE_CONFIG["boundary_loss_weight"])
    assert abs(float(out["loss"]) - float(ref)) / float(ref) < LOSS_TOL, (float(out["loss"]), float(ref))
