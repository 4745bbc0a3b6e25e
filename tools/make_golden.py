#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own modules on CPU.

Runs ONLY in the build container (needs /root/reference); the outputs are small data
fixtures under tests/golden/ (inputs' seeds / tensors and expected outputs -- no
reference source).  Weights are never stored: they are regenerated from
``oracle.detweights`` (pure function of key name + shape + seed).

Reference assembly:
  flow  = flow_model.MaskedDiffWithXvec(encoder=<vendored cosyvoice ConformerEncoder>,
          length_regulator=<vendored InterpolateRegulator>,
          decoder=flow_model.ConditionalCFM(estimator=modules.ConditionalDecoder('gelu')))
          with modules.GELU.approximate switched to 'none' (diffusers' default) for the
          "vendored" variant, and flow_model.build_flow_model() untouched for the "twin".
  llm   = cosyvoice.llm.llm.TransformerLM over vendored Conformer/Transformer encoders
          (two empty import shims: torchaudio, omegaconf -- import-time only).
  joint = llm_flow_model.JointLLMFlowModel ; LoRA via lora.apply_lora_to_model.

usage: python tools/make_golden.py [--only tiny|full|ops|train]
"""
import argparse
import json
import math
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cosyvoice_flow_finetune"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

from transformers import Qwen2ForCausalLM  # noqa: F401  (must precede the shims)

sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
_om = types.ModuleType("omegaconf")
_om.DictConfig = type("DictConfig", (dict,), {})
sys.modules.setdefault("omegaconf", _om)

import flow_model as ref_flow_model          # noqa: E402
import llm_flow_model as ref_joint           # noqa: E402
import lora as ref_lora                      # noqa: E402
import modules as ref_modules                # noqa: E402
from cosyvoice.flow.length_regulator import InterpolateRegulator as VInterp          # noqa: E402
from cosyvoice.llm.llm import TransformerLM                                            # noqa: E402
from cosyvoice.transformer.encoder import ConformerEncoder as VConformer             # noqa: E402
from cosyvoice.transformer.encoder import TransformerEncoder as VTransformer         # noqa: E402
from cosyvoice.transformer.attention import RelPositionMultiHeadedAttention as VRelMHA  # noqa: E402
from cosyvoice.transformer.label_smoothing_loss import LabelSmoothingLoss             # noqa: E402
from cosyvoice.transformer.convolution import ConvolutionModule as VConvModule        # noqa: E402

from oracle.detweights import det_state_dict                                           # noqa: E402
from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch, cfm_draws     # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")

FLOW_TARGETS = ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'w_1', 'w_2']   # config.py:207-216
LLM_TARGETS = ['linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2']              # config.py:195-204

TINY_FLOW = dict(input_size=128, vocab_size=64, encoder_attention_heads=2, encoder_linear_units=256,
                 encoder_num_blocks=2, decoder_channels=(64, 64), decoder_attention_head_dim=64,
                 decoder_n_blocks=1, decoder_num_mid_blocks=2, decoder_num_heads=2)
TINY_LLM = dict(text_in=64, d=128, heads=2, ff=256, text_blocks=2, llm_blocks=2, text_vocab=100, speech_vocab=50)
FULL_LLM = dict(text_in=512, d=1024, heads=16, ff=4096, text_blocks=6, llm_blocks=14, text_vocab=51866, speech_vocab=4096)


def build_ref_flow(variant: str, **kw):
    """variant 'vendored' | 'twin'."""
    m = ref_flow_model.build_flow_model(pretrained_path=None, device='cpu', **kw)
    if variant == 'vendored':
        d = kw.get('input_size', 512)
        m.encoder = VConformer(
            input_size=d, output_size=d, attention_heads=kw.get('encoder_attention_heads', 8),
            linear_units=kw.get('encoder_linear_units', 2048), num_blocks=kw.get('encoder_num_blocks', 6),
            dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1, normalize_before=True,
            input_layer='linear', pos_enc_layer_type='rel_pos_espnet', selfattention_layer_type='rel_selfattn',
            use_cnn_module=False, macaron_style=False)
        m.length_regulator = VInterp(channels=80, sampling_ratios=(1, 1, 1, 1), out_channels=80, groups=1)
        for mod in m.modules():
            if isinstance(mod, ref_modules.GELU):
                mod.approximate = 'none'
    return m


def build_ref_llm(c):
    te = VConformer(input_size=c['text_in'], output_size=c['d'], attention_heads=c['heads'], linear_units=c['ff'],
                    num_blocks=c['text_blocks'], dropout_rate=0.1, positional_dropout_rate=0.1,
                    attention_dropout_rate=0.0, normalize_before=True, input_layer='linear',
                    pos_enc_layer_type='rel_pos_espnet', selfattention_layer_type='rel_selfattn',
                    use_cnn_module=False, macaron_style=False, use_dynamic_chunk=False,
                    use_dynamic_left_chunk=False, static_chunk_size=1)
    lm = VTransformer(input_size=c['d'], output_size=c['d'], attention_heads=c['heads'], linear_units=c['ff'],
                      num_blocks=c['llm_blocks'], dropout_rate=0.1, positional_dropout_rate=0.1,
                      attention_dropout_rate=0.0, input_layer='linear_legacy', pos_enc_layer_type='rel_pos_espnet',
                      selfattention_layer_type='rel_selfattn', static_chunk_size=1)
    return TransformerLM(text_encoder_input_size=c['text_in'], llm_input_size=c['d'], llm_output_size=c['d'],
                         text_token_size=c['text_vocab'], speech_token_size=c['speech_vocab'], text_encoder=te,
                         llm=lm, sampling=None, length_normalized_loss=True, lsm_weight=0.0, spk_embed_dim=192)


def wrap_and_fill(model, r, alpha, targets, seed):
    stats = ref_lora.apply_lora_to_model(model, r=r, lora_alpha=alpha, lora_dropout=0.0, target_modules=targets)
    spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(det_state_dict(spec, seed), strict=True)
    model.eval()
    return spec, stats


def lora_grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if 'lora_' in n and p.grad is not None}


def npz_save(path, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


def run_flow(model, batch, seed):
    jm = ref_joint.JointLLMFlowModel(nn.Identity(), model, 'flow_only')
    model.zero_grad(set_to_none=True)
    torch.manual_seed(seed)
    out = jm(batch, torch.device('cpu'))
    out['loss'].backward()
    return out['loss'].detach(), lora_grads(model)


def gen_tiny():
    # ---------------- flow ----------------
    batch = synth_batch([24, 17], text_lens=[7, 5], token_lens=[13, 9], seed=11, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(2, 24, seed=77)
    res = {}
    for variant in ('vendored', 'twin'):
        torch.manual_seed(0)
        m = build_ref_flow(variant, **TINY_FLOW)
        spec, stats = wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
        loss, grads = run_flow(m, batch, seed=77)
        res[variant] = (loss, grads, spec, stats)
        print(f"tiny flow {variant}: loss={loss.item():.8f}  lora tensors={len(grads)} stats={stats}")
    # intermediates of the vendored variant through hooks
    torch.manual_seed(0)
    m = build_ref_flow('vendored', **TINY_FLOW)
    spec, _ = wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
    cap = {}
    m.encoder.register_forward_hook(lambda mod, i, o: cap.__setitem__('h_enc', o[0].detach()))
    m.length_regulator.register_forward_hook(lambda mod, i, o: cap.__setitem__('mu', o[0].detach()))
    m.decoder.estimator.register_forward_hook(
        lambda mod, i, o: cap.update(pred=o.detach(), y=i[0].detach(), t=i[3].detach()))
    loss2, _ = run_flow(m, batch, seed=77)
    assert torch.equal(loss2, res['vendored'][0])
    arr = dict(loss_vendored=res['vendored'][0], loss_twin=res['twin'][0], **{f"in_{k}": v for k, v in batch.items()},
               draw_t_raw=draws['t_raw'], draw_z=draws['z'], draw_cfg_rand=draws['cfg_rand'],
               h_enc=cap['h_enc'], mu=cap['mu'], pred=cap['pred'], y=cap['y'], t=cap['t'])
    for k, g in res['vendored'][1].items():
        arr[f"grad_vendored/{k}"] = g
    for k, g in res['twin'][1].items():
        arr[f"grad_twin/{k}"] = g
    npz_save(os.path.join(GOLD, "flow_tiny.npz"), **arr)
    meta = dict(flow=dict(build=TINY_FLOW, lora=dict(r=4, alpha=8, targets=FLOW_TARGETS), weight_seed=3, draw_seed=77,
                          spec=[[k, list(s)] for k, s in spec], stats=res['vendored'][3],
                          batch=dict(feat_lens=[24, 17], text_lens=[7, 5], token_lens=[13, 9], seed=11,
                                     text_vocab=100, speech_vocab=50)))

    # ---------------- llm ----------------
    torch.manual_seed(0)
    llm = build_ref_llm(TINY_LLM)
    lspec, lstats = wrap_and_fill(llm, r=4, alpha=8, targets=LLM_TARGETS, seed=5)
    jm = ref_joint.JointLLMFlowModel(llm, nn.Identity(), 'llm_only')
    cap = {}
    llm.llm_decoder.register_forward_hook(lambda mod, i, o: cap.__setitem__('logits', o.detach()))
    llm.text_encoder_affine_layer.register_forward_hook(lambda mod, i, o: cap.__setitem__('text_enc', o.detach()))
    out = jm(batch, torch.device('cpu'))
    out['loss'].backward()
    lg = lora_grads(llm)
    print(f"tiny llm: loss={out['loss'].item():.8f} acc={out['llm_acc'].item():.6f} stats={lstats}")
    arr = dict(loss=out['loss'].detach(), acc=out['llm_acc'], logits=cap['logits'], text_enc=cap['text_enc'],
               **{f"in_{k}": v for k, v in batch.items()})
    for k, g in lg.items():
        arr[f"grad/{k}"] = g
    npz_save(os.path.join(GOLD, "llm_tiny.npz"), **arr)
    meta['llm'] = dict(build=TINY_LLM, lora=dict(r=4, alpha=8, targets=LLM_TARGETS), weight_seed=5,
                       spec=[[k, list(s)] for k, s in lspec], stats=lstats)

    # ---------------- merged export contract (lora.py:284-323) ----------------
    torch.manual_seed(0)
    m = build_ref_flow('vendored', **TINY_FLOW)
    base_keys = sorted(m.state_dict().keys())
    wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
    merged = ref_lora.get_merged_state_dict(m)
    assert sorted(merged.keys()) == base_keys
    m2 = build_ref_flow('vendored', **TINY_FLOW)
    m2.load_state_dict(merged, strict=True)
    m2.eval()
    loss_m, _ = run_flow_nograd(m2, batch, 77)
    print("merged-model loss", loss_m.item(), "vs lora", res['vendored'][0].item())
    meta['flow']['base_keys'] = base_keys
    meta['flow']['merged_loss'] = float(loss_m)
    pick = ['decoder.estimator.mid_blocks.0.1.0.attn1.to_q.weight', 'encoder.encoders.1.feed_forward.w_2.weight']
    npz_save(os.path.join(GOLD, "flow_tiny_merged.npz"), **{k: merged[k] for k in pick})
    with open(os.path.join(GOLD, "tiny_meta.json"), "w") as f:
        json.dump(meta, f, indent=0)


def gen_oddT():
    """Odd padded T_max (VERDICT round 2 item 4 / SURVEY Appendix C): the U-Net's stride-2 down path gives ceil(T/2) frames and the
    transposed-conv up path 2*ceil(T/2) = T + 1, cropped back to T (cosyvoice/flow/decoder.py:256, 276 == modules.py:1049, 1080).
    Ragged B = 2, T in {25, 18} (T_max = 25 odd, and an odd + an even utterance), all LoRA gradients, both numerics variants."""
    batch = synth_batch([25, 18], text_lens=[6, 5], token_lens=[14, 10], seed=21, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(2, 25, seed=78)
    arr = dict(**{f"in_{k}": v for k, v in batch.items()}, draw_t_raw=draws['t_raw'], draw_z=draws['z'], draw_cfg_rand=draws['cfg_rand'])
    for variant in ('vendored', 'twin'):
        torch.manual_seed(0)
        m = build_ref_flow(variant, **TINY_FLOW)
        wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
        cap = {}
        if variant == 'vendored':
            m.decoder.estimator.register_forward_hook(lambda mod, i, o: cap.update(pred=o.detach()))
        loss, grads = run_flow(m, batch, seed=78)
        print(f"odd-T flow {variant}: loss={loss.item():.8f}  lora tensors={len(grads)}")
        arr[f"loss_{variant}"] = loss
        if cap:
            arr["pred"] = cap["pred"]
        for k, g in grads.items():
            arr[f"grad_{variant}/{k}"] = g
    npz_save(os.path.join(GOLD, "flow_oddT.npz"), **arr)
    with open(os.path.join(GOLD, "flow_oddT_meta.json"), "w") as f:
        json.dump(dict(feat_lens=[25, 18], text_lens=[6, 5], token_lens=[14, 10], seed=21, draw_seed=78, text_vocab=100,
                       speech_vocab=50), f)


def gen_pos():
    """LLM tiny with the reference's DEFAULT target list (lora.py:155-166: target_modules=None), which wraps linear_pos:
    the gradient then also flows into the projected positional encoding (rel-pos attention's p operand)."""
    batch = synth_batch([24, 17], text_lens=[7, 5], token_lens=[13, 9], seed=11, text_vocab=100, speech_vocab=50)
    torch.manual_seed(0)
    llm = build_ref_llm(TINY_LLM)
    lspec, lstats = wrap_and_fill(llm, r=4, alpha=8, targets=None, seed=5)
    assert any('linear_pos.lora_A' in k for k, _ in lspec)
    jm = ref_joint.JointLLMFlowModel(llm, nn.Identity(), 'llm_only')
    out = jm(batch, torch.device('cpu'))
    out['loss'].backward()
    lg = lora_grads(llm)
    print(f"tiny llm, default targets: loss={out['loss'].item():.8f} acc={out['llm_acc'].item():.6f} stats={lstats}")
    arr = dict(loss=out['loss'].detach(), acc=out['llm_acc'], **{f"in_{k}": v for k, v in batch.items()})
    for k, g in lg.items():
        arr[f"grad/{k}"] = g
    npz_save(os.path.join(GOLD, "llm_tiny_pos.npz"), **arr)
    with open(os.path.join(GOLD, "tiny_pos_meta.json"), "w") as f:
        json.dump(dict(llm=dict(build=TINY_LLM, lora=dict(r=4, alpha=8, targets=None), weight_seed=5,
                                spec=[[k, list(s)] for k, s in lspec], stats=lstats)), f, indent=0)


def gen_sampler():
    """CFM sampler fixture: reference ConditionalCFM.forward on the tiny vendored flow model."""
    torch.manual_seed(0)
    m = build_ref_flow('vendored', **TINY_FLOW)
    wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
    m.eval()
    g = torch.Generator().manual_seed(41)
    T = 44
    mu = torch.randn(1, 80, T, generator=g) * 0.5
    spks = torch.randn(1, 80, generator=g) * 0.3
    cond = torch.zeros(1, 80, T)
    cond[:, :, :10] = torch.randn(1, 80, 10, generator=g)          # a 10-frame prompt
    mask = torch.ones(1, 1, T)
    z = torch.randn(1, 80, T, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: z.clone().to(t.dtype)    # pin the sampler's initial noise
    try:
        out, cache = m.decoder(mu.clone(), mask, n_timesteps=5, temperature=1.0, spks=spks, cond=cond, prompt_len=10)
    finally:
        torch.randn_like = orig
    print(f"sampler: out mean {out.mean().item():.6f} std {out.std().item():.6f} cache {tuple(cache.shape)}")
    npz_save(os.path.join(GOLD, "sampler_tiny.npz"), mu=mu, spks=spks, cond=cond, mask=mask, z=z, out=out, cache=cache)


def gen_inference():
    """Flow inference entries (second half): MaskedDiffWithXvec.inference (prompt + target tokens, the
    head/mid/tail length regulator, length-dependent step count) and inference_like_training on the tiny flow model,
    initial noise pinned."""
    torch.manual_seed(0)
    m = build_ref_flow('vendored', **TINY_FLOW)
    wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
    m.eval()
    g = torch.Generator().manual_seed(58)
    token = torch.randint(0, 64, (1, 46), generator=g)
    ptoken = torch.randint(0, 64, (1, 12), generator=g)
    pfeat = torch.randn(1, 20, 80, generator=g)
    emb = torch.randn(1, 192, generator=g)
    mel2 = int(46 / 50 * 22050 / 256)
    z = torch.randn(1, 80, 20 + mel2, generator=g)
    z2 = torch.randn(1, 80, 52, generator=g)
    orig = torch.randn_like
    out = {}
    try:
        torch.randn_like = lambda t, *a, **k: z.clone().to(t.dtype)
        mel, cache = m.inference(token, torch.tensor([46]), ptoken, torch.tensor([12]), pfeat, torch.tensor([20]), emb)
        out.update(inf_mel=mel, inf_cache=cache)
        torch.randn_like = lambda t, *a, **k: z2.clone().to(t.dtype)
        out['ilt_mel'] = m.inference_like_training(token, torch.tensor([46]), 52, emb, prompt_feat=pfeat, prompt_len=9, n_timesteps=10)
        out['ilt_mel_noprompt'] = m.inference_like_training(token, torch.tensor([46]), torch.tensor([52]), emb, n_timesteps=4)
    finally:
        torch.randn_like = orig
    print("inference:", {k: tuple(v.shape) for k, v in out.items()})
    npz_save(os.path.join(GOLD, "flow_inference_tiny.npz"), token=token, prompt_token=ptoken, prompt_feat=pfeat, embedding=emb,
             z=z, z2=z2, **out)


def gen_data():
    """On-disk data path fixture: a small shard in the schema prepare_joint_data.py writes
    (275-284, 365-372) + what the reference's dataset.py makes of it under fixed seeds."""
    import random
    import pandas as pd
    import pyarrow as pa
    import pyarrow.parquet as pq
    import dataset as ref_ds
    d = os.path.join(GOLD, "data_shard")
    os.makedirs(os.path.join(d, "parquet"), exist_ok=True)
    g = torch.Generator().manual_seed(99)
    rows = []
    for i, T in enumerate([61, 40, 75, 33, 52, 58]):
        mel = torch.randn(T, 80, generator=g) * 2 - 6
        n_tok = int(T * 50 * 256 / 22050)
        emb = torch.randn(192, generator=g).tolist()
        rows.append({'utt': f'utt{i}', 'text': f'text {i}', 'text_token': torch.randint(0, 500, (5 + i,), generator=g).tolist(),
                     'speech_token': torch.randint(0, 4096, (n_tok,), generator=g).tolist(),
                     'speech_feat': mel.flatten().tolist(), 'speech_feat_shape': (T, 80),
                     'utt_embedding': emb, 'spk_embedding': emb})
    rows[3]['text_token'] = None                                   # one utterance without text
    pq.write_table(pa.Table.from_pandas(pd.DataFrame(rows)), os.path.join(d, "parquet", "data_000000.parquet"))
    with open(os.path.join(d, "data.list"), "w", encoding="utf-8") as f:
        f.write("D:\\out\\parquet\\data_000000.parquet\n")   # a path from the machine that wrote it
    ref_ds.ANTI_LEAKAGE_CONFIG = {'cross_sample_enabled': True, 'cross_sample_prob': 0.5}
    ds = ref_ds.FlowFinetuneDataset(d, augmentation=True)
    ds.cross_sample_enabled, ds.cross_sample_prob = True, 0.5
    arr = {}
    random.seed(7)
    torch.manual_seed(7)
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        for k, v in it.items():
            if v is not None:
                arr[f"item{i}/{k}"] = v
    for name, idx, lim in (("b0", [0, 1, 2], 50), ("b1", [3, 4], 50), ("b2", [0, 5], 1000)):
        out = ref_ds.collate_fn([{k: (v.clone() if torch.is_tensor(v) else v) for k, v in items[i].items()} for i in idx], lim)
        for k, v in out.items():
            arr[f"{name}/{k}"] = v
    ds2 = ref_ds.FlowFinetuneDataset(d, augmentation=False)
    ds2.cross_sample_enabled = False
    for k, v in ds2[2].items():
        if v is not None:
            arr[f"plain2/{k}"] = v
    npz_save(os.path.join(GOLD, "data_path.npz"), **arr)


def gen_prompt():
    """Anti-leakage flow-only training path: the reference's MaskedDiffWithXvec.forward with prompts
    on the tiny LoRA-wrapped flow model, under random.seed / torch.manual_seed; the per-utterance decisions it drew are
    recovered from the compute_loss call and stored with the loss and the LoRA gradients."""
    import random
    import flow_model as ref_fm
    cfg_al = {'silence_padding_enabled': True, 'silence_min_tokens': 5, 'silence_max_tokens': 10, 'silence_mel_value': -11.5,
              'dynamic_prompt_enabled': True, 'prompt_min_ratio': 0.10, 'prompt_max_ratio': 0.30,
              'prompt_dropout_enabled': True, 'prompt_dropout_prob': 0.25, 'boundary_loss_enabled': True, 'boundary_frames': 25,
              'boundary_loss_weight': 3.0, 'cross_sample_enabled': True, 'cross_sample_prob': 0.5,
              'text_blinding_enabled': True, 'text_blinding_prob': 0.7}
    old = (ref_fm.ANTI_LEAKAGE_CONFIG, ref_fm.NO_PROMPT_TRAINING_CONFIG)
    ref_fm.ANTI_LEAKAGE_CONFIG, ref_fm.NO_PROMPT_TRAINING_CONFIG = cfg_al, {'enabled': False}
    try:
        batch = synth_batch([44, 37, 29, 40], text_lens=[7, 5, 6, 4], token_lens=[24, 20, 15, 21], seed=21, text_vocab=100, speech_vocab=50)
        g = torch.Generator().manual_seed(8)
        batch['cross_sample_mel'] = torch.randn(4, 9, 80, generator=g) * 2 - 6
        batch['cross_sample_mel_len'] = torch.tensor([9, 0, 6, 9])
        torch.manual_seed(0)
        m = build_ref_flow('vendored', **TINY_FLOW)
        spec, _ = wrap_and_fill(m, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
        m.eval()
        cap = {}
        orig = m.decoder.compute_loss

        def spy(x1, mask, mu, spks=None, cond=None, prompt_lens=None):
            cap.update(mu=mu.detach().clone(), cond=cond.detach().clone(), prompt_lens=list(prompt_lens))
            return orig(x1, mask, mu, spks, cond=cond, prompt_lens=prompt_lens)
        m.decoder.compute_loss = spy
        random.seed(123)
        torch.manual_seed(77)
        out = m(batch, torch.device('cpu'))
        out['loss'].backward()
        grads = lora_grads(m)
        draws = cfm_draws(4, 44, seed=77)
    finally:
        ref_fm.ANTI_LEAKAGE_CONFIG, ref_fm.NO_PROMPT_TRAINING_CONFIG = old
    print(f"prompt path: loss={out['loss'].item():.8f} prompt_lens={cap['prompt_lens']}")
    arr = dict(loss=out['loss'].detach(), cond=cap['cond'], mu=cap['mu'], prompt_lens=torch.tensor(cap['prompt_lens']),
               draw_t_raw=draws['t_raw'], draw_z=draws['z'], draw_cfg_rand=draws['cfg_rand'], **{f"in_{k}": v for k, v in batch.items()})
    for k, gr in grads.items():
        arr[f"grad/{k}"] = gr
    npz_save(os.path.join(GOLD, "flow_prompt_tiny.npz"), **arr)
    with open(os.path.join(GOLD, "flow_prompt_meta.json"), "w") as f:
        json.dump(dict(anti_leakage=cfg_al, random_seed=123, draw_seed=77), f, indent=1)


def run_flow_nograd(model, batch, seed):
    jm = ref_joint.JointLLMFlowModel(nn.Identity(), model, 'flow_only')
    torch.manual_seed(seed)
    with torch.no_grad():
        out = jm(batch, torch.device('cpu'))
    return out['loss'], None


def grad_summary(grads):
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    keys = sorted(grads.keys())
    pick = [keys[0], keys[len(keys) // 2], keys[-1]]
    return dict(total_norm=tot, picks={k: float(grads[k].double().norm()) for k in pick},
                abs_sum={k: float(grads[k].double().abs().sum()) for k in pick})


def gen_full():
    """Full CosyVoice-300M dims, BASELINE shapes (T=500), r=16: scalars only."""
    res = {}
    cases = {
        "uniform_T500_B2": dict(feat_lens=[500, 500], text_lens=[40, 40]),
        "ragged_T500_B2": dict(feat_lens=[500, 437], text_lens=[40, 33]),
    }
    torch.manual_seed(0)
    flow = build_ref_flow('vendored')
    fspec, fstats = wrap_and_fill(flow, r=16, alpha=32, targets=FLOW_TARGETS, seed=1)
    print("full flow stats", fstats)
    res['flow_stats'] = fstats
    res['flow_lora'] = dict(r=16, alpha=32, targets=FLOW_TARGETS, weight_seed=1)
    for name, c in cases.items():
        batch = synth_batch(c['feat_lens'], text_lens=c['text_lens'], seed=1234)
        loss, grads = run_flow(flow, batch, seed=4321)
        res[f"flow/{name}"] = dict(loss=float(loss), draw_seed=4321, batch_seed=1234, **c, grads=grad_summary(grads))
        print(name, "flow loss", float(loss), res[f"flow/{name}"]['grads']['total_norm'])
    res['flow_spec_len'] = len(fspec)
    del flow
    torch.manual_seed(0)
    llm = build_ref_llm(FULL_LLM)
    lspec, lstats = wrap_and_fill(llm, r=16, alpha=32, targets=LLM_TARGETS, seed=2)
    print("full llm stats", lstats)
    res['llm_stats'] = lstats
    res['llm_lora'] = dict(r=16, alpha=32, targets=LLM_TARGETS, weight_seed=2)
    jm = ref_joint.JointLLMFlowModel(llm, nn.Identity(), 'llm_only')
    for name, c in cases.items():
        batch = synth_batch(c['feat_lens'], text_lens=c['text_lens'], seed=1234)
        llm.zero_grad(set_to_none=True)
        out = jm(batch, torch.device('cpu'))
        out['loss'].backward()
        grads = lora_grads(llm)
        res[f"llm/{name}"] = dict(loss=float(out['loss']), acc=float(out['llm_acc']), batch_seed=1234, **c,
                                  grads=grad_summary(grads))
        print(name, "llm loss", float(out['loss']), float(out['llm_acc']), res[f"llm/{name}"]['grads']['total_norm'])
    with open(os.path.join(GOLD, "full_scalars.json"), "w") as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(GOLD, "full_spec.json"), "w") as f:
        json.dump(dict(flow=[[k, list(s)] for k, s in fspec], llm=[[k, list(s)] for k, s in lspec]), f)


def pick_grad_tensors(grads, n=6):
    """a handful of whole LoRA gradient tensors, spread over the model (first / middle / last adapters, A and B)"""
    keys = sorted(grads)
    idx = sorted({0, 1, len(keys) // 2, len(keys) // 2 + 1, len(keys) - 2, len(keys) - 1})[:n]
    return {keys[i]: grads[keys[i]] for i in idx}


def gen_full_grads():
    """Whole LoRA gradient tensors of a few adapters at full CosyVoice-300M size, for the bf16 BACKWARD checks (the
    scalar fixtures of gen_full pin losses and gradient norms only), and the BASELINE configs[4] shape (T = 1000,
    r = 64, B = 2): losses, accuracy, gradient norms and the same picked tensors."""
    arrs, meta = {}, {}
    for tag, T, r, feat_lens, text_lens in (("t500_r16", 500, 16, [500, 437], [40, 33]),
                                            ("t1000_r64", 1000, 64, [1000, 871], [40, 33])):
        batch = synth_batch(feat_lens, text_lens=text_lens, seed=1234)
        torch.manual_seed(0)
        flow = build_ref_flow('vendored')
        fspec, fstats = wrap_and_fill(flow, r=r, alpha=2 * r, targets=FLOW_TARGETS, seed=1)
        loss, grads = run_flow(flow, batch, seed=4321)
        meta[f"flow/{tag}"] = dict(loss=float(loss), draw_seed=4321, batch_seed=1234, feat_lens=feat_lens, text_lens=text_lens,
                                   r=r, alpha=2 * r, weight_seed=1, grads=grad_summary(grads), stats=fstats)
        for k, g in pick_grad_tensors(grads).items():
            arrs[f"flow/{tag}/{k}"] = g
        print(tag, "flow loss", float(loss), meta[f"flow/{tag}"]['grads']['total_norm'])
        del flow, grads
        torch.manual_seed(0)
        llm = build_ref_llm(FULL_LLM)
        lspec, lstats = wrap_and_fill(llm, r=r, alpha=2 * r, targets=LLM_TARGETS, seed=2)
        jm = ref_joint.JointLLMFlowModel(llm, nn.Identity(), 'llm_only')
        out = jm(batch, torch.device('cpu'))
        out['loss'].backward()
        grads = lora_grads(llm)
        meta[f"llm/{tag}"] = dict(loss=float(out['loss']), acc=float(out['llm_acc']), batch_seed=1234, feat_lens=feat_lens,
                                  text_lens=text_lens, r=r, alpha=2 * r, weight_seed=2, grads=grad_summary(grads), stats=lstats)
        for k, g in pick_grad_tensors(grads).items():
            arrs[f"llm/{tag}/{k}"] = g
        print(tag, "llm loss", float(out['loss']), float(out['llm_acc']), meta[f"llm/{tag}"]['grads']['total_norm'])
        del llm, jm, grads
    npz_save(os.path.join(GOLD, "full_grads.npz"), **arrs)
    with open(os.path.join(GOLD, "full_grads_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_ops():
    """Op-level known answers from the reference's own classes."""
    g = torch.Generator().manual_seed(5)
    arr = {}
    # rel_shift (attention.py:225-247)
    att = VRelMHA(2, 128, 0.0)
    x = torch.randn(2, 2, 7, 13, generator=g)
    arr['rel_shift_in'] = x
    arr['rel_shift_out'] = att.rel_shift(x)
    # espnet rel-pos table (embedding.py:201-302)
    from cosyvoice.transformer.embedding import EspnetRelPositionalEncoding
    pe = EspnetRelPositionalEncoding(16, 0.0)
    xs, pos = pe(torch.zeros(1, 9, 16))
    arr['relpos_table_L9_d16'] = pos
    # LabelSmoothingLoss(smoothing 0, normalize_length) (label_smoothing_loss.py:68-96)
    crit = LabelSmoothingLoss(size=11, padding_idx=-1, smoothing=0.0, normalize_length=True)
    lg = torch.randn(3, 5, 11, generator=g)
    tg = torch.randint(0, 11, (3, 5), generator=g)
    tg[0, :2] = -1
    tg[2, 4] = -1
    arr['ce_logits'], arr['ce_target'], arr['ce_loss'] = lg, tg, crit(lg, tg)
    from cosyvoice.utils.common import th_accuracy
    arr['ce_acc'] = th_accuracy(lg.view(-1, 11), tg, ignore_label=-1)
    # SinusoidalPosEmb scale=1000 (modules.py:20-42)
    t = torch.tensor([0.0, 0.137, 0.5, 0.999])
    arr['sin_t'], arr['sin_emb'] = t, ref_modules.SinusoidalPosEmb(320)(t)
    # LoRALinear fwd (lora.py:64-76)
    lin = nn.Linear(24, 40)
    ll = ref_lora.LoRALinear(lin, r=4, lora_alpha=8, lora_dropout=0.0)
    spec = [(k, tuple(v.shape)) for k, v in ll.state_dict().items()]
    ll.load_state_dict(det_state_dict(spec, 9))
    xx = torch.randn(5, 24, generator=g)
    arr['lora_x'], arr['lora_y'] = xx, ll(xx)
    # mask helpers (utils.py:20-109)
    import utils as ref_utils
    arr['pad_mask_5_3_2'] = ref_utils.make_pad_mask(torch.tensor([5, 3, 2]))
    arr['chunk_mask_6_1'] = ref_utils.subsequent_chunk_mask(6, 1)
    arr['mask_bias'] = ref_utils.mask_to_bias(torch.tensor([[True, False, True]]), torch.float32)
    # Conformer ConvolutionModule (convolution.py:24-145), layer_norm variant, k=15
    cm = VConvModule(32, 15, nn.SiLU(), 'layer_norm', causal=False)
    cspec = [(k, tuple(v.shape)) for k, v in cm.state_dict().items()]
    cm.load_state_dict(det_state_dict(cspec, 13))
    cm.eval()
    cx = torch.randn(2, 21, 32, generator=g)
    mp = ~ref_utils.make_pad_mask(torch.tensor([21, 16])).unsqueeze(1)
    arr['convmod_x'], arr['convmod_mask'] = cx, mp
    arr['convmod_y'] = cm(cx.clone(), mp)[0]
    # LoRAConv1d (lora.py:79-131)
    c1 = nn.Conv1d(12, 20, 1)
    lc = ref_lora.LoRAConv1d(c1, r=4, lora_alpha=8, lora_dropout=0.0)
    sspec = [(k, tuple(v.shape)) for k, v in lc.state_dict().items()]
    lc.load_state_dict(det_state_dict(sspec, 17))
    xc = torch.randn(2, 12, 9, generator=g)
    arr['loraconv_x'], arr['loraconv_y'] = xc, lc(xc)
    # CFM identities (flow_matching.py:173-181)
    npz_save(os.path.join(GOLD, "ops.npz"), **arr)
    with open(os.path.join(GOLD, "ops_meta.json"), "w") as f:
        json.dump(dict(lora_spec=[[k, list(s)] for k, s in spec], lora_seed=9,
                       convmod_spec=[[k, list(s)] for k, s in cspec], convmod_seed=13,
                       loraconv_spec=[[k, list(s)] for k, s in sspec], loraconv_seed=17), f)


def gen_lsm():
    """LabelSmoothingLoss with smoothing > 0 (label_smoothing_loss.py:68-96): loss and d loss / d logits, both normalisations."""
    g = torch.Generator().manual_seed(11)
    arr = {}
    lg = torch.randn(3, 6, 13, generator=g)
    tg = torch.randint(0, 13, (3, 6), generator=g)
    tg[0, :2] = -1
    tg[1, 5] = -1
    arr['logits'], arr['target'] = lg, tg
    for eps in (0.1, 0.3, 1.0):
        for nl in (True, False):
            x = lg.clone().requires_grad_(True)
            loss = LabelSmoothingLoss(size=13, padding_idx=-1, smoothing=eps, normalize_length=nl)(x, tg)
            loss.backward()
            tag = f"eps{eps}_{'tok' if nl else 'batch'}"
            arr['loss_' + tag], arr['grad_' + tag] = loss.detach(), x.grad
    np.savez(os.path.join(GOLD, "lsm.npz"), **{k: v.detach().numpy() for k, v in arr.items()})
    print("wrote lsm.npz")


def gen_train():
    """BASELINE configs[0] plumbing reference: 8 synthetic pairs, LoRA r=4, fp32 CPU, 2 epochs,
    batch 1, accumulate 2, AdamW + warmup-cosine LambdaLR + clip 1.0 (train_joint.py:198-226,
    349-360 arithmetic under a hand-rolled loop: pytorch_lightning is not installable offline)."""
    torch.manual_seed(0)
    flow = build_ref_flow('vendored', **TINY_FLOW)
    wrap_and_fill(flow, r=4, alpha=8, targets=FLOW_TARGETS, seed=3)
    llm = build_ref_llm(TINY_LLM)
    wrap_and_fill(llm, r=4, alpha=8, targets=LLM_TARGETS, seed=5)
    jm = ref_joint.JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0)
    jm.eval()
    hp = dict(lr=2e-3, min_lr=1e-6, warmup=2, wd=0.01, accum=2, clip=1.0, epochs=2, n=8)
    params = [p for p in jm.parameters() if p.requires_grad]
    names = [n for n, p in jm.named_parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=hp['lr'], weight_decay=hp['wd'], betas=(0.9, 0.999))
    total_steps = hp['epochs'] * math.ceil(hp['n'] / hp['accum'])

    def lr_lambda(step):
        if step < hp['warmup']:
            return step / max(1, hp['warmup'])
        progress = (step - hp['warmup']) / max(1, total_steps - hp['warmup'])
        return max(hp['min_lr'] / hp['lr'], 0.5 * (1 + torch.cos(torch.tensor(progress * 3.14159)).item()))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
    lens = [(20 + 2 * i, 5 + (i % 3), 9 + i) for i in range(hp['n'])]
    log = []
    for ep in range(hp['epochs']):
        for i in range(hp['n']):
            T, Lx, Lt = lens[i]
            batch = synth_batch([T], text_lens=[Lx], token_lens=[Lt], seed=100 + i, text_vocab=100, speech_vocab=50)
            torch.manual_seed(1000 * ep + i)
            out = jm(batch, torch.device('cpu'))
            (out['loss'] / hp['accum']).backward()
            rec = dict(epoch=ep, idx=i, loss=float(out['loss']), llm_loss=float(out['llm_loss']),
                       flow_loss=float(out['flow_loss']), llm_acc=float(out['llm_acc']))
            if (i + 1) % hp['accum'] == 0:
                gn = float(torch.nn.utils.clip_grad_norm_(params, hp['clip']))
                rec['grad_norm'] = gn
                rec['lr'] = opt.param_groups[0]['lr']
                opt.step()
                sched.step()
                opt.zero_grad(set_to_none=True)
            log.append(rec)
            print(rec)
    final = {n: p.detach().clone() for n, p in zip(names, params)}
    npz_save(os.path.join(GOLD, "train_tiny_final.npz"), **final)
    with open(os.path.join(GOLD, "train_tiny_log.json"), "w") as f:
        json.dump(dict(hp=hp, lens=lens, total_steps=total_steps, log=log), f, indent=0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if a.only in ("all", "ops"):
        gen_ops()
    if a.only in ("all", "lsm"):
        gen_lsm()
    if a.only in ("all", "tiny"):
        gen_tiny()
    if a.only in ("all", "pos"):
        gen_pos()
    if a.only in ("all", "oddT"):
        gen_oddT()
    if a.only in ("all", "fullgrads"):
        gen_full_grads()
    if a.only in ("all", "sampler"):
        gen_sampler()
    if a.only in ("all", "data"):
        gen_data()
    if a.only in ("all", "prompt"):
        gen_prompt()
    if a.only in ("all", "inference"):
        gen_inference()
    if a.only in ("all", "train"):
        gen_train()
    if a.only in ("all", "full"):
        gen_full()
