"""CPU oracle: a functional fp32 restatement of the reference's joint LLM+Flow LoRA step.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use
it, and only as the checker / the reported CPU baseline -- never as the thing shipped.

The oracle works directly on a reference-format ``state_dict`` (the exact key names the
reference's modules produce, including ``<path>.original_layer.*`` / ``<path>.lora_A`` /
``<path>.lora_B`` after ``lora.apply_lora_to_model``), so it also pins the state-dict
key contract.

Parity pin: ``tests/golden/*.npz`` were produced by ``tools/make_golden.py`` which runs
the *reference's own modules* (imported from /root/reference in the build container) on
the same weights / inputs / CFM draws; ``tests/test_oracle_golden.py`` checks every
function below against those outputs.  Every function cites the reference code it
restates (paths relative to /root/reference/cosyvoice_flow_finetune/).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

IGNORE_ID = -1  # cosyvoice/utils/common.py:25


@dataclass
class OracleConfig:
    """Numerics switches (SURVEY.md section 8c "deviations").

    Defaults = the *vendored* path that ``train_joint.py`` really runs
    (cosyvoice.flow.* + matcha + diffusers): x*sqrt(d) in the rel-pos encoding,
    erf-GELU in the estimator feed-forward, LayerNorm eps 1e-12 in encoder layers.
    ``OracleConfig.twin()`` gives the importable self-contained twin
    (flow_model.py + modules.py): no xscale, tanh-GELU, eps 1e-5.
    """
    flow_xscale: bool = True          # embedding.py:267-270 vs modules.py:414-420
    gelu_approximate: str = "none"    # diffusers GELU default vs modules.py:132-139
    flow_enc_ln_eps: float = 1e-12    # encoder_layer.py:145-155 vs modules.py:664-675
    est_head_dim: int = 64            # flow_model.py:651 decoder_attention_head_dim
    est_groups: int = 8               # matcha decoder.py Block1D groups=8
    flow_lora_scale: float = 2.0      # lora_alpha / r  (config.py:207-216: 32/16)
    llm_lora_scale: float = 2.0       # config.py:195-204: 16/8
    mel_mean: float = -6.0            # config.py:241
    mel_std: float = 2.0              # config.py:242
    sigma_min: float = 1e-6           # flow_model.py:702
    training_cfg_rate: float = 0.2    # flow_model.py:704
    llm_ln_eps: float = 1e-12         # encoder_layer.py:50-51
    llm_xscale: bool = True
    llm_text_causal: bool = True      # static_chunk_size=1 (upstream cosyvoice.yaml)
    llm_causal: bool = True
    speech_token_size: int = 4096

    @staticmethod
    def twin(**kw) -> "OracleConfig":
        return OracleConfig(flow_xscale=False, gelu_approximate="tanh", flow_enc_ln_eps=1e-5, **kw)


# ---------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------

def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """utils.py:20-42 == cosyvoice/utils/mask.py:237-265. True at padded positions."""
    max_len = max_len if max_len > 0 else int(lengths.max().item())
    rng = torch.arange(0, max_len, dtype=torch.int64, device=lengths.device)
    return rng.unsqueeze(0) >= lengths.unsqueeze(-1)


def mask_to_bias(mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """utils.py:103-109 / cosyvoice/utils/common.py:160-168: (1-m) * -1e10."""
    assert mask.dtype == torch.bool
    return (1.0 - mask.to(dtype)) * -1.0e10


def lora_linear(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, scale: float) -> torch.Tensor:
    """lora.py:64-76 (dropout off): W x + b + scale * B(A x); plain Linear if not wrapped."""
    if f"{p}.lora_A" in sd:
        y = F.linear(x, sd[f"{p}.original_layer.weight"], sd.get(f"{p}.original_layer.bias"))
        u = F.linear(x, sd[f"{p}.lora_A"])
        return y + F.linear(u, sd[f"{p}.lora_B"]) * scale
    return F.linear(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"))


def _count(sd, prefix: str) -> int:
    """Number of consecutive integer children under ``prefix.<i>.``."""
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def rel_pos_table(L: int, d: int, dtype=torch.float32) -> torch.Tensor:
    """embedding.py:222-255 + 272-302 restated: row m holds the encoding of relative
    position (L-1-m), m in [0, 2L-1)."""
    pos = (L - 1 - torch.arange(0, 2 * L - 1, dtype=torch.float32)).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * L - 1, d, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0).to(dtype)


def rel_shift(x: torch.Tensor) -> torch.Tensor:
    """attention.py:225-247 as its index law: y[..., i, j] = x[..., i, L-1-i+j]."""
    L = x.size(2)
    i = torch.arange(L).unsqueeze(1)
    j = torch.arange(L).unsqueeze(0)
    idx = (L - 1 - i + j).expand(x.size(0), x.size(1), L, L)
    return torch.gather(x, 3, idx)


def rel_mha(sd, p: str, x: torch.Tensor, mask: torch.Tensor, pos_emb: torch.Tensor, scale: float) -> torch.Tensor:
    """attention.py:276-330 + 82-127 (RelPositionMultiHeadedAttention, no cache)."""
    B, L, _ = x.shape
    u, v_ = sd[f"{p}.pos_bias_u"], sd[f"{p}.pos_bias_v"]
    h, dk = u.shape
    q = lora_linear(sd, f"{p}.linear_q", x, scale).view(B, L, h, dk)
    k = lora_linear(sd, f"{p}.linear_k", x, scale).view(B, L, h, dk).transpose(1, 2)
    v = lora_linear(sd, f"{p}.linear_v", x, scale).view(B, L, h, dk).transpose(1, 2)
    pp = lora_linear(sd, f"{p}.linear_pos", pos_emb, scale).view(1, -1, h, dk).transpose(1, 2)
    qu = (q + u).transpose(1, 2)
    qv = (q + v_).transpose(1, 2)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd = torch.matmul(qv, pp.transpose(-2, -1))
    if ac.shape != bd.shape:
        bd = rel_shift(bd)
    scores = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    scores = scores.masked_fill(m, -float("inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, L, h * dk)
    return lora_linear(sd, f"{p}.linear_out", o, scale)


_ACTS = {"relu": F.relu, "swish": F.silu}


def encoder(sd, p: str, xs: torch.Tensor, lens: torch.Tensor, *, kind: str, causal: bool,
            ln_eps: float, xscale: bool, scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """cosyvoice/transformer/encoder.py:111-170 (BaseEncoder.forward) with
    LinearNoSubsampling (subsampling.py:69-113) or LegacyLinearNoSubsampling (338-383,
    detected by ``kind``), EspnetRelPositionalEncoding, static chunk (causal) masking
    (mask.py:223-236) and Transformer / Conformer(no macaron, no cnn) layers
    (encoder_layer.py:90-106 / 202-236).  Dropout off."""
    T = xs.size(1)
    masks = ~make_pad_mask(lens, T).unsqueeze(1)                       # (B,1,T)
    x = lora_linear(sd, f"{p}.embed.out.0", xs, scale)
    d = x.size(-1)
    x = F.layer_norm(x, (d,), sd[f"{p}.embed.out.1.weight"], sd[f"{p}.embed.out.1.bias"], 1e-5)
    if kind == "transformer":                                           # linear_legacy
        x = F.relu(x)
    if xscale:
        x = x * math.sqrt(d)
    pos_emb = rel_pos_table(T, d, x.dtype)
    if causal:
        tri = torch.tril(torch.ones(T, T, dtype=torch.bool)).unsqueeze(0)
        cm = masks & tri
    else:
        cm = masks
    dead = cm.sum(dim=-1) == 0
    if dead.any():
        cm = cm.clone()
        cm[dead] = True
    act = F.relu if kind == "transformer" else F.silu
    n1, n2 = ("norm1", "norm2") if kind == "transformer" else ("norm_mha", "norm_ff")
    for i in range(_count(sd, f"{p}.encoders")):
        lp = f"{p}.encoders.{i}"
        if f"{lp}.feed_forward_macaron.w_1.weight" in sd or f"{lp}.feed_forward_macaron.w_1.original_layer.weight" in sd:
            r = x
            y = F.layer_norm(x, (d,), sd[f"{lp}.norm_ff_macaron.weight"], sd[f"{lp}.norm_ff_macaron.bias"], ln_eps)
            y = lora_linear(sd, f"{lp}.feed_forward_macaron.w_2", act(lora_linear(sd, f"{lp}.feed_forward_macaron.w_1", y, scale)), scale)
            x = r + 0.5 * y
            ff_scale = 0.5
        else:
            ff_scale = 1.0
        r = x
        y = F.layer_norm(x, (d,), sd[f"{lp}.{n1}.weight"], sd[f"{lp}.{n1}.bias"], ln_eps)
        x = r + rel_mha(sd, f"{lp}.self_attn", y, cm, pos_emb, scale)
        r = x
        y = F.layer_norm(x, (d,), sd[f"{lp}.{n2}.weight"], sd[f"{lp}.{n2}.bias"], ln_eps)
        y = lora_linear(sd, f"{lp}.feed_forward.w_2", act(lora_linear(sd, f"{lp}.feed_forward.w_1", y, scale)), scale)
        x = r + ff_scale * y
    x = F.layer_norm(x, (d,), sd[f"{p}.after_norm.weight"], sd[f"{p}.after_norm.bias"], 1e-5)
    return x, masks


# ---------------------------------------------------------------------------------
# Flow: estimator (U-Net1D) + CFM loss
# ---------------------------------------------------------------------------------

def sinusoidal_pos_emb(t: torch.Tensor, dim: int, scale: float = 1000.0) -> torch.Tensor:
    """matcha/models/components/decoder.py:14-32 == modules.py:20-42."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = scale * t.unsqueeze(1) * e.unsqueeze(0)
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv1d(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"), stride=stride, padding=padding)


def block1d(sd, p: str, x, mask, groups: int):
    """matcha decoder.py:35-47 == modules.py:60-73."""
    y = _conv(sd, f"{p}.block.0", x * mask, padding=1)
    y = F.group_norm(y, groups, sd[f"{p}.block.1.weight"], sd[f"{p}.block.1.bias"], 1e-5)
    return F.mish(y) * mask


def resnet1d(sd, p: str, x, mask, temb, groups: int):
    """matcha decoder.py:50-66 == modules.py:76-94."""
    h = block1d(sd, f"{p}.block1", x, mask, groups)
    h = h + F.linear(F.mish(temb), sd[f"{p}.mlp.1.weight"], sd[f"{p}.mlp.1.bias"]).unsqueeze(-1)
    h = block1d(sd, f"{p}.block2", h, mask, groups)
    return h + _conv(sd, f"{p}.res_conv", x * mask)


def basic_transformer_block(sd, p: str, x, bias, cfg: OracleConfig):
    """matcha transformer.py:255-316 over diffusers Attention/FeedForward(GELU)
    == modules.py:227-293, 349-375.  x (B,T,C); bias (B,T,T) additive."""
    B, T, C = x.shape
    s = cfg.flow_lora_scale
    y = F.layer_norm(x, (C,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)
    q = lora_linear(sd, f"{p}.attn1.to_q", y, s)
    k = lora_linear(sd, f"{p}.attn1.to_k", y, s)
    v = lora_linear(sd, f"{p}.attn1.to_v", y, s)
    inner = q.size(-1)
    h = inner // cfg.est_head_dim
    q, k, v = (t.view(B, T, h, cfg.est_head_dim).transpose(1, 2) for t in (q, k, v))
    sim = torch.matmul(q, k.transpose(-2, -1)) * (cfg.est_head_dim ** -0.5)
    sim = sim + bias.unsqueeze(1)
    o = torch.matmul(sim.softmax(dim=-1), v).transpose(1, 2).reshape(B, T, inner)
    x = x + lora_linear(sd, f"{p}.attn1.to_out.0", o, s)
    y = F.layer_norm(x, (C,), sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], 1e-5)
    y = F.gelu(lora_linear(sd, f"{p}.ff.net.0.proj", y, s), approximate=cfg.gelu_approximate)
    return x + lora_linear(sd, f"{p}.ff.net.2", y, s)


def estimator(sd, p: str, x, mask, mu, t, spks, cond, cfg: OracleConfig, prompt_len: int = 0):
    """cosyvoice/flow/decoder.py:210-291 == modules.py:998-1106.  x,mu,cond (B,80,T); mask (B,1,T); t (B,); spks (B,80).
    prompt_len > 0: the twin's prompt-isolation bias (modules.py:844-879, 1033-1042) -- block-diagonal -inf at the
    split rescaled to each U-Net level."""
    T_full = x.shape[-1]
    g = cfg.est_groups
    in_ch = sd[f"{p}.time_mlp.linear_1.weight"].shape[1]
    temb = sinusoidal_pos_emb(t, in_ch).to(t.dtype)
    temb = F.linear(temb, sd[f"{p}.time_mlp.linear_1.weight"], sd[f"{p}.time_mlp.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd[f"{p}.time_mlp.linear_2.weight"], sd[f"{p}.time_mlp.linear_2.bias"])
    x = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, x.shape[-1]), cond], dim=1)

    def tblocks(x, m, bp):
        xt = x.transpose(1, 2).contiguous()
        bias = mask_to_bias(m.bool().expand(-1, xt.size(1), -1), xt.dtype)
        if prompt_len > 0:
            n = xt.size(1)
            sp = max(1, int(prompt_len * (n / T_full)))
            if sp < n:
                iso = torch.zeros(1, n, n, dtype=xt.dtype)
                iso[:, sp:, :sp] = float("-inf")
                iso[:, :sp, sp:] = float("-inf")
                bias = bias + iso
        for j in range(_count(sd, bp)):
            xt = basic_transformer_block(sd, f"{bp}.{j}", xt, bias, cfg)
        return xt.transpose(1, 2).contiguous()

    hiddens, masks = [], [mask]
    for i in range(_count(sd, f"{p}.down_blocks")):
        bp = f"{p}.down_blocks.{i}"
        m = masks[-1]
        x = resnet1d(sd, f"{bp}.0", x, m, temb, g)
        x = tblocks(x, m, f"{bp}.1")
        hiddens.append(x)
        if f"{bp}.2.conv.weight" in sd:
            x = _conv(sd, f"{bp}.2.conv", x * m, stride=2, padding=1)
        else:
            x = _conv(sd, f"{bp}.2", x * m, padding=1)
        masks.append(m[:, :, ::2])
    masks = masks[:-1]
    mm = masks[-1]
    for i in range(_count(sd, f"{p}.mid_blocks")):
        bp = f"{p}.mid_blocks.{i}"
        x = resnet1d(sd, f"{bp}.0", x, mm, temb, g)
        x = tblocks(x, mm, f"{bp}.1")
    for i in range(_count(sd, f"{p}.up_blocks")):
        bp = f"{p}.up_blocks.{i}"
        m = masks.pop()
        skip = hiddens.pop()
        x = torch.cat([x[:, :, :skip.shape[-1]], skip], dim=1)
        x = resnet1d(sd, f"{bp}.0", x, m, temb, g)
        x = tblocks(x, m, f"{bp}.1")
        if f"{bp}.2.conv.weight" in sd:
            x = F.conv_transpose1d(x * m, sd[f"{bp}.2.conv.weight"], sd[f"{bp}.2.conv.bias"], stride=2, padding=1)
        else:
            x = _conv(sd, f"{bp}.2", x * m, padding=1)
    x = block1d(sd, f"{p}.final_block", x, m, g)
    return _conv(sd, f"{p}.final_proj", x * m) * mask


def _regulator_stack(sd, p: str, x: torch.Tensor):
    """The conv stack of InterpolateRegulator (length_regulator.py:29-42): [Conv1d k3, GroupNorm(1), Mish] x n, Conv1d k1."""
    n = 0
    while f"{p}.model.{n + 1}.weight" in sd and sd[f"{p}.model.{n}.weight"].dim() == 3 and sd[f"{p}.model.{n + 1}.weight"].dim() == 1:
        x = _conv(sd, f"{p}.model.{n}", x, padding=1)
        x = F.group_norm(x, 1, sd[f"{p}.model.{n + 1}.weight"], sd[f"{p}.model.{n + 1}.bias"], 1e-5)
        x = F.mish(x)
        n += 3
    return _conv(sd, f"{p}.model.{n}", x)


def length_regulator(sd, p: str, x: torch.Tensor, ylens: torch.Tensor):
    """cosyvoice/flow/length_regulator.py:44-50 == modules.py:817-821."""
    mask = (~make_pad_mask(ylens)).to(x).unsqueeze(-1)
    x = F.interpolate(x.transpose(1, 2).contiguous(), size=int(ylens.max()), mode="linear")
    return _regulator_stack(sd, p, x).transpose(1, 2).contiguous() * mask


def length_regulator_inference(sd, p: str, x1, x2, mel_len1: int, mel_len2: int, input_frame_rate: int = 50):
    """length_regulator.py:52-70 == modules.py:823-838: prompt part and target part interpolated separately, the target's
    first / last 20 tokens at the nominal rate and the middle taking up the slack, so the prompt/target seam is exact."""
    up = lambda t, n: F.interpolate(t.transpose(1, 2).contiguous(), size=n, mode="linear")
    if x2.shape[1] > 40:
        e = int(20 / input_frame_rate * 22050 / 256)
        x2 = torch.cat([up(x2[:, :20], e), up(x2[:, 20:-20], mel_len2 - 2 * e), up(x2[:, -20:], e)], dim=2)
    else:
        x2 = up(x2, mel_len2)
    x = torch.cat([up(x1, mel_len1), x2], dim=2) if x1.shape[1] != 0 else x2
    return _regulator_stack(sd, p, x).transpose(1, 2).contiguous()


def ode_steps_for(n_frames: int) -> int:
    """flow_model.py:525-536: Euler steps by sequence length."""
    return 20 if n_frames > 500 else 15 if n_frames > 300 else 10


def cfm_prepare(x1, t_raw, z, sigma_min: float, t_scheduler: str = 'cosine'):
    """cosyvoice/flow/flow_matching.py:173-181 == flow_model.py:143-155.
    t_raw = rand(B,1,1); returns t (B,1,1), y, u."""
    t = 1 - torch.cos(t_raw * 0.5 * torch.pi) if t_scheduler == 'cosine' else t_raw
    y = (1 - (1 - sigma_min) * t) * z + t * x1
    u = x1 - (1 - sigma_min) * z
    return t, y, u


def cfm_sample(sd, p: str, z, mu, mask, spks, cond, n_timesteps: int, cfg: OracleConfig, inference_cfg_rate: float = 0.7):
    """flow_model.py:74-135 (ConditionalCFM.forward + solve_euler, no cache): cosine t-span, batch-of-2 CFG Euler steps.
    z,mu,cond (1,80,T); mask (1,1,T); spks (1,80) -> (1,80,T) fp32."""
    t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=mu.dtype)
    t_span = 1 - torch.cos(t_span * 0.5 * 3.14159265359)
    x, t, dt = z, t_span[0].unsqueeze(0), t_span[1] - t_span[0]
    for step in range(1, len(t_span)):
        x_in = torch.cat([x, x], 0)
        mask_in = torch.cat([mask, mask], 0)
        mu_in = torch.cat([mu, torch.zeros_like(mu)], 0)
        spks_in = torch.cat([spks, torch.zeros_like(spks)], 0)
        cond_in = torch.cat([cond, torch.zeros_like(cond)], 0)
        d = estimator(sd, p, x_in, mask_in, mu_in, t.expand(2), spks_in, cond_in, cfg)
        x = x + dt * ((1.0 + inference_cfg_rate) * d[:1] - inference_cfg_rate * d[1:])
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return x.float()


def _flow_encode(sd, token, token_len, embedding, cfg: OracleConfig):
    emb = lora_linear(sd, "spk_embed_affine_layer", F.normalize(embedding.float(), dim=1), cfg.flow_lora_scale)
    tmask = (~make_pad_mask(token_len)).float().unsqueeze(-1)
    tok = F.embedding(torch.clamp(token, min=0), sd["input_embedding.weight"]) * tmask
    h, _ = encoder(sd, "encoder", tok, token_len, kind="conformer", causal=False, ln_eps=cfg.flow_enc_ln_eps,
                   xscale=cfg.flow_xscale, scale=cfg.flow_lora_scale)
    return lora_linear(sd, "encoder_proj", h, cfg.flow_lora_scale), emb


def flow_inference(sd, token, prompt_token, prompt_feat, embedding, z, cfg: OracleConfig, input_frame_rate: int = 50):
    """flow_model.py:474-551 (MaskedDiffWithXvec.inference, batch 1, no flow cache): prompt + target tokens through the
    encoder, head/mid/tail length regulation, prompt mel as conditioning, Euler steps by length; returns the target part
    (1,80,mel_len2) and the cache (1,80,prompt+34,2) of flow_model.py:84-93.  z (1,80,T): the sampler's initial noise."""
    n1, n2 = prompt_token.shape[1], token.shape[1]
    tok = torch.cat([prompt_token, token], dim=1)
    h, emb = _flow_encode(sd, tok, torch.tensor([n1 + n2]), embedding, cfg)
    mel1, mel2 = prompt_feat.shape[1], int(n2 / input_frame_rate * 22050 / 256)
    mu = length_regulator_inference(sd, "length_regulator", h[:, :n1], h[:, n1:], mel1, mel2, input_frame_rate).transpose(1, 2)
    cond = torch.zeros(1, 80, mel1 + mel2)
    cond[:, :, :mel1] = prompt_feat.transpose(1, 2)
    cache = torch.stack([torch.cat([z[:, :, :mel1], z[:, :, -34:]], dim=2), torch.cat([mu[:, :, :mel1], mu[:, :, -34:]], dim=2)], dim=-1)
    out = cfm_sample(sd, "decoder.estimator", z, mu, torch.ones(1, 1, mel1 + mel2), emb, cond, ode_steps_for(mel1 + mel2), cfg)
    return out[:, :, mel1:], cache


def flow_inference_like_training(sd, token, feat_len: int, embedding, z, cfg: OracleConfig, prompt_feat=None,
                                 prompt_len: int = 0, n_timesteps: int = 10):
    """flow_model.py:553-638: whole token sequence, plain length regulation to feat_len, optional prompt conditioning."""
    h, emb = _flow_encode(sd, token, torch.tensor([token.shape[1]]), embedding, cfg)
    mu = length_regulator(sd, "length_regulator", h, torch.tensor([feat_len])).transpose(1, 2)
    cond = torch.zeros(1, 80, feat_len)
    if prompt_feat is not None and prompt_len > 0:
        n = min(prompt_len, prompt_feat.shape[1], feat_len)
        cond[:, :, :n] = prompt_feat[:, :n].transpose(1, 2)
    if n_timesteps is None or n_timesteps == 10:
        n_timesteps = ode_steps_for(feat_len)
    return cfm_sample(sd, "decoder.estimator", z, mu, torch.ones(1, 1, feat_len), emb, cond, n_timesteps, cfg)


def flow_forward(sd, batch, draws, cfg: OracleConfig, return_all: bool = False):
    """llm_flow_model.py:181-229 (_forward_flow, no-prompt) ->
    ConditionalCFM.compute_loss (flow_matching.py:154-193).
    draws = dict(t_raw (B,1,1), z (B,80,T), cfg_rand (B,)) in the reference's draw order."""
    s = cfg.flow_lora_scale
    token, token_len = batch["speech_token"], batch["speech_token_len"]
    feat = (batch["speech_feat"].float() - cfg.mel_mean) / cfg.mel_std
    feat_len = batch["speech_feat_len"]
    emb = F.normalize(batch["embedding"].float(), dim=1)
    emb = lora_linear(sd, "spk_embed_affine_layer", emb, s)
    tmask = (~make_pad_mask(token_len)).float().unsqueeze(-1)
    tok = F.embedding(torch.clamp(token, min=0), sd["input_embedding.weight"]) * tmask
    h_enc, _ = encoder(sd, "encoder", tok, token_len, kind="conformer", causal=False,
                       ln_eps=cfg.flow_enc_ln_eps, xscale=cfg.flow_xscale, scale=s)
    h = lora_linear(sd, "encoder_proj", h_enc, s)
    mu = length_regulator(sd, "length_regulator", h, feat_len)                 # (B,T,80)
    x1 = feat.transpose(1, 2).contiguous()
    mask = (~make_pad_mask(feat_len)).to(mu).unsqueeze(1)
    mu_t = mu.transpose(1, 2).contiguous()
    cond = torch.zeros_like(x1)
    t, y, u = cfm_prepare(x1, draws["t_raw"], draws["z"], cfg.sigma_min)
    cm = (draws["cfg_rand"] > cfg.training_cfg_rate).to(x1.dtype)
    pred = estimator(sd, "decoder.estimator", y, mask, mu_t * cm.view(-1, 1, 1), t.view(-1),
                     emb * cm.view(-1, 1), cond * cm.view(-1, 1, 1), cfg)
    loss = F.mse_loss(pred * mask, u * mask, reduction="sum") / (torch.sum(mask) * u.shape[1])
    if return_all:
        return dict(loss=loss, h_enc=h_enc, mu=mu, y=y, u=u, pred=pred, t=t.view(-1))
    return loss


def flow_forward_prompt(sd, batch, draws, cfg: OracleConfig, plan, boundary_frames: int, boundary_weight: float,
                        silence_value: float = -11.5, return_all: bool = False):
    """flow_model.py:248-400 + 137-204 (flow-only training with a mel prompt and the anti-leakage strategies), given the
    per-utterance decisions `plan` = [{total, copy, cross, silence, blind}] the reference draws from `random`."""
    s = cfg.flow_lora_scale
    token, token_len = batch["speech_token"], batch["speech_token_len"]
    feat = (batch["speech_feat"].float() - cfg.mel_mean) / cfg.mel_std
    feat_len = batch["speech_feat_len"]
    emb = F.normalize(batch["embedding"].float(), dim=1)
    emb = lora_linear(sd, "spk_embed_affine_layer", emb, s)
    tmask = (~make_pad_mask(token_len)).float().unsqueeze(-1)
    tok = F.embedding(torch.clamp(token, min=0), sd["input_embedding.weight"]) * tmask
    h_enc, _ = encoder(sd, "encoder", tok, token_len, kind="conformer", causal=False,
                       ln_eps=cfg.flow_enc_ln_eps, xscale=cfg.flow_xscale, scale=s)
    h = lora_linear(sd, "encoder_proj", h_enc, s)
    mu = length_regulator(sd, "length_regulator", h, feat_len)                 # (B,T,80)
    conds = torch.zeros_like(feat)
    keep = torch.ones_like(mu[:, :, :1])
    cross = None
    if "cross_sample_mel" in batch:
        cross = (batch["cross_sample_mel"].float() - cfg.mel_mean) / cfg.mel_std
    for i, pl in enumerate(plan):
        c = pl["copy"]
        if c > 0:
            conds[i, :c] = cross[i, :c] if pl["cross"] else feat[i, :c]
            if pl["silence"] > 0:
                conds[i, c:c + pl["silence"]] = (silence_value - cfg.mel_mean) / cfg.mel_std
            if pl["blind"]:
                keep[i, :c] = 0.0
    mu = mu * keep
    x1 = feat.transpose(1, 2).contiguous()
    mask = (~make_pad_mask(feat_len)).to(mu).unsqueeze(1)
    t, y, u = cfm_prepare(x1, draws["t_raw"], draws["z"], cfg.sigma_min)
    cm = (draws["cfg_rand"] > cfg.training_cfg_rate).to(x1.dtype)
    prompt_lens = [pl["total"] for pl in plan]
    pred = estimator(sd, "decoder.estimator", y, mask, mu.transpose(1, 2) * cm.view(-1, 1, 1), t.view(-1),
                     emb * cm.view(-1, 1), conds.transpose(1, 2) * cm.view(-1, 1, 1), cfg, prompt_len=max(prompt_lens))
    loss_mask = mask.clone()
    for i, pl in enumerate(prompt_lens):
        if pl > 0:
            loss_mask[i, :, :pl] = 0
            loss_mask[i, :, pl:min(pl + boundary_frames, loss_mask.shape[2])] = boundary_weight
    diff = (pred - u) * loss_mask
    loss = (diff ** 2).sum() / (torch.sum(loss_mask) * u.shape[1])
    return dict(loss=loss, cond=conds.transpose(1, 2), mu=mu.transpose(1, 2)) if return_all else loss


# ---------------------------------------------------------------------------------
# LLM
# ---------------------------------------------------------------------------------

def ce_ignore(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """label_smoothing_loss.py:68-96 with smoothing 0 + normalize_length:
    -sum(log_softmax(x)[tgt]) over non-ignored / #non-ignored."""
    V = logits.size(-1)
    x = logits.view(-1, V)
    tgt = target.view(-1)
    ign = tgt == IGNORE_ID
    total = tgt.numel() - int(ign.sum())
    lp = torch.log_softmax(x, dim=1)
    nll = -lp.gather(1, tgt.masked_fill(ign, 0).unsqueeze(1)).squeeze(1)
    return nll.masked_fill(ign, 0).sum() / total


def ce_label_smoothing(logits: torch.Tensor, target: torch.Tensor, smoothing: float, normalize_length: bool = True) -> torch.Tensor:
    """label_smoothing_loss.py:68-96 in full: true_dist = smoothing/(V-1) off the target, 1-smoothing on it; KL summed
    over the non-ignored rows (a zero target probability contributes 0), divided by #non-ignored tokens
    (normalize_length) or by the batch size."""
    V = logits.size(-1)
    x = logits.reshape(-1, V)
    tgt = target.reshape(-1)
    ign = tgt == IGNORE_ID
    total = tgt.numel() - int(ign.sum())
    td = torch.full_like(x, smoothing / (V - 1))
    td.scatter_(1, tgt.masked_fill(ign, 0).unsqueeze(1), 1.0 - smoothing)
    lp = torch.log_softmax(x, dim=1)
    kl = torch.where(td > 0, td * (td.clamp_min(1e-45).log() - lp), torch.zeros_like(lp))
    return kl.masked_fill(ign.unsqueeze(1), 0).sum() / (total if normalize_length else logits.size(0))


def th_accuracy(logits2d: torch.Tensor, target: torch.Tensor, ignore_label: int = IGNORE_ID) -> torch.Tensor:
    """cosyvoice/utils/common.py:78-97."""
    pred = logits2d.view(target.size(0), target.size(1), logits2d.size(1)).argmax(2)
    m = target != ignore_label
    return (torch.sum(pred.masked_select(m) == target.masked_select(m)) / torch.sum(m)).detach()


def build_lm_target(text_token_len, speech_token, speech_token_len, speech_token_size: int):
    """llm_flow_model.py:129-139."""
    tg = [torch.tensor([IGNORE_ID] * (2 + int(text_token_len[i])) +
                       speech_token[i, :int(speech_token_len[i])].tolist() + [speech_token_size])
          for i in range(speech_token.size(0))]
    return torch.nn.utils.rnn.pad_sequence(tg, batch_first=True, padding_value=IGNORE_ID)


def llm_forward(sd, batch, cfg: OracleConfig, return_all: bool = False):
    """llm_flow_model.py:109-179 (_forward_llm, no-prompt) over
    cosyvoice/llm/llm.py:78-95 (encode, pad_unpad_sequence)."""
    s = cfg.llm_lora_scale
    text, text_len = batch["text_token"], batch["text_token_len"]
    sp, sp_len = batch["speech_token"], batch["speech_token_len"]
    V = cfg.speech_token_size
    lm_target = build_lm_target(text_len, sp, sp_len, V)
    temb = F.embedding(text, sd["text_embedding.weight"])
    enc, enc_mask = encoder(sd, "text_encoder", temb, text_len, kind="conformer", causal=cfg.llm_text_causal,
                            ln_eps=cfg.llm_ln_eps, xscale=cfg.llm_xscale, scale=s)
    enc_len = enc_mask.squeeze(1).sum(1)
    enc = lora_linear(sd, "text_encoder_affine_layer", enc, s)
    emb = F.normalize(batch["embedding"].float(), dim=1)
    emb = lora_linear(sd, "spk_embed_affine_layer", emb, s).unsqueeze(1)
    sos = sd["llm_embedding.weight"][0].reshape(1, -1)
    task = sd["llm_embedding.weight"][1].reshape(1, -1)
    semb = F.embedding(sp, sd["speech_embedding.weight"])
    seqs = [torch.cat([sos, emb[i], enc[i, :int(enc_len[i])], task, semb[i, :int(sp_len[i])]], dim=0)
            for i in range(text.size(0))]
    lens = torch.tensor([x.size(0) for x in seqs], dtype=torch.int32)
    lm_in = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=IGNORE_ID)
    out, _ = encoder(sd, "llm", lm_in, lens, kind="transformer", causal=cfg.llm_causal,
                     ln_eps=cfg.llm_ln_eps, xscale=cfg.llm_xscale, scale=s)
    logits = lora_linear(sd, "llm_decoder", out, s)
    loss = ce_ignore(logits, lm_target)
    acc = th_accuracy(logits.view(-1, V + 1), lm_target)
    if return_all:
        return dict(loss=loss, acc=acc, logits=logits, lm_target=lm_target, text_enc=enc, lm_in=lm_in)
    return loss, acc


def joint_forward(sd_llm, sd_flow, batch, draws, cfg: OracleConfig, mode: str = "joint",
                  llm_w: float = 1.0, flow_w: float = 1.0):
    """llm_flow_model.py:77-107."""
    out = {}
    if mode in ("joint", "llm_only"):
        l, a = llm_forward(sd_llm, batch, cfg)
        out["llm_loss"], out["llm_acc"] = l * llm_w, a
    if mode in ("joint", "flow_only"):
        out["flow_loss"] = flow_forward(sd_flow, batch, draws, cfg) * flow_w
    out["loss"] = {"joint": lambda: out["llm_loss"] + out["flow_loss"],
                   "llm_only": lambda: out["llm_loss"], "flow_only": lambda: out["flow_loss"]}[mode]()
    return out


# ---------------------------------------------------------------------------------
# standalone ops named by north_star but not executed by the 300M config
# ---------------------------------------------------------------------------------

def conformer_conv_module(sd, p: str, x: torch.Tensor, mask_pad: torch.Tensor, act=F.silu, causal=False):
    """cosyvoice/transformer/convolution.py:86-145 (layer_norm variant, no cache).
    x (B,T,C); mask_pad (B,1,T) bool."""
    x = x.transpose(1, 2).masked_fill(~mask_pad, 0.0)
    K = sd[f"{p}.depthwise_conv.weight"].shape[-1]
    if causal:
        x = F.pad(x, (K - 1, 0))
    x = F.glu(_conv(sd, f"{p}.pointwise_conv1", x), dim=1)
    C = x.size(1)
    x = F.conv1d(x, sd[f"{p}.depthwise_conv.weight"], sd.get(f"{p}.depthwise_conv.bias"),
                 padding=0 if causal else (K - 1) // 2, groups=C)
    x = act(F.layer_norm(x.transpose(1, 2), (C,), sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-5)).transpose(1, 2)
    x = _conv(sd, f"{p}.pointwise_conv2", x).masked_fill(~mask_pad, 0.0)
    return x.transpose(1, 2)


def lora_grads(loss: torch.Tensor, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(names, gs)}
