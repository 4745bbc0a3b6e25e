"""Flow model (speech tokens -> mel, conditional flow matching) on the HIP path.

Same classes / constructor arguments / state-dict keys as the reference's flow_model.py
(``ConditionalCFM`` flow_model.py:50-204, ``MaskedDiffWithXvec`` 207-246,
``build_flow_model`` 641-767) == vendored cosyvoice/flow/{flow,flow_matching}.py; only the
training hot path (``compute_loss`` and the no-prompt forward that llm_flow_model.py:181-229
drives) plus the SURVEY.md section 8(f) items: the Euler sampler (rank 3) and the flow-only
anti-leakage training path (rank 4).
"""
from __future__ import annotations

import os
import random
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from .config import ANTI_LEAKAGE_CONFIG, MEL_MEAN, MEL_STD, NO_PROMPT_TRAINING_CONFIG
from .hipops import functional as HF
from .modules import (ConditionalDecoder, InterpolateRegulator, Numerics, RelPosEncoder, hip_linear, to_len)


class ConditionalCFM(nn.Module):
    def __init__(self, in_channels: int, n_spks: int = 1, spk_emb_dim: int = 64, sigma_min: float = 1e-6,
                 t_scheduler: str = 'cosine', training_cfg_rate: float = 0.2, inference_cfg_rate: float = 0.7,
                 estimator: Optional[nn.Module] = None):
        super().__init__()
        self.in_channels, self.n_spks, self.spk_emb_dim = in_channels, n_spks, spk_emb_dim
        self.sigma_min, self.t_scheduler = sigma_min, t_scheduler
        self.training_cfg_rate, self.inference_cfg_rate = training_cfg_rate, inference_cfg_rate
        self.estimator = estimator

    @staticmethod
    def make_draws(B: int, T: int, device, n_mels: int = 80) -> Dict[str, torch.Tensor]:
        """The reference's three draws, in its order (flow_matching.py:175-186), on `device`."""
        return {"t_raw": torch.rand([B, 1, 1], device=device), "z": torch.randn(B, n_mels, T, device=device),
                "cfg_rand": torch.rand(B, device=device)}

    def compute_loss_cl(self, feat, mu, spk, length, B: int, T: int, num: Numerics, draws=None,
                        mel_mean: float = 0.0, mel_std: float = 1.0, cond=None, prompt_lens=None, t_true=None):
        """Channel-last hot path.  feat [B,T,80] fp32 (raw log-mel if mel_mean/std given), mu [B*T,80],
        spk [B,80], length int32 [B] -> scalar loss (flow_matching.py:154-193).
        prompt_lens (list of B ints, flow_model.py:164-202): prompt-isolation attention in the estimator (split at the
        batch's longest prompt), loss weight 0 on each sample's prompt frames and `boundary_loss_weight` on the
        `boundary_frames` after them."""
        dev = mu.device
        if draws is None:
            draws = self.make_draws(B, T, dev)
        t_raw = draws["t_raw"].reshape(B).to(dev, torch.float32)
        z = draws["z"].to(dev, torch.float32).transpose(1, 2)   # reference layout (B,80,T) -> channel-last
        keep = (draws["cfg_rand"].to(dev) > self.training_cfg_rate).to(torch.float32) if self.training_cfg_rate > 0 \
            else torch.ones(B, device=dev)
        xin, u, t = HF.cfm_prepare(mu, spk, feat.to(dev, torch.float32).contiguous(), z.contiguous(), t_raw, keep, B, T,
                                   mel_mean, mel_std, self.sigma_min, cond, self.t_scheduler == 'cosine')
        if prompt_lens is not None and len(prompt_lens) > 0:
            self.estimator.prompt_isolation_len = int(max(prompt_lens))
            self.estimator.prompt_isolation_enabled = True
        else:
            self.estimator.prompt_isolation_len = 0
        try:
            pred = self.estimator.forward_cl(xin, t, B, T, length, num.gelu, t_true)
        finally:
            self.estimator.prompt_isolation_len = 0
        if prompt_lens is None:
            denom = (length.sum() * pred.shape[1]).to(torch.float32)
            return HF.masked_mse(pred, u, length, denom, B, T), xin
        # loss_mask of flow_model.py:179-196, built on the host from lengths: padding mask, prompt frames 0, then the
        # boundary window overwritten with its weight (as the reference does, also where it reaches past the length)
        lens = length.detach().cpu().tolist()
        w = torch.zeros(B, T, dtype=torch.float32)
        bf, bw = ANTI_LEAKAGE_CONFIG.get('boundary_frames', 15), ANTI_LEAKAGE_CONFIG.get('boundary_loss_weight', 3.0)
        for i, pl in enumerate(prompt_lens):
            w[i, :lens[i]] = 1.0
            if pl > 0:
                w[i, :pl] = 0.0
                if ANTI_LEAKAGE_CONFIG.get('boundary_loss_enabled', True):
                    w[i, pl:min(pl + bf, T)] = bw
        w = w.reshape(-1).to(dev)
        denom = (w.sum() * pred.shape[1]).clamp_min(1e-20)
        return HF.masked_mse(pred, u, None, denom, B, T, weight=w), xin

    @torch.no_grad()
    def forward(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None, prompt_len=0, cache=None,
                noise: Optional[torch.Tensor] = None, num: Optional[Numerics] = None):
        """CFM sampler (flow_model.py:74-98): cosine t-span Euler ODE with classifier-free guidance.
        mu,cond (1,80,T), mask (1,1,T), spks (1,80) -> (mel (1,80,T) fp32, new_cache).  `noise` (optional) injects the
        initial z (the reference draws `randn_like(mu) * temperature`)."""
        z = (torch.randn_like(mu) if noise is None else noise.to(mu.device, mu.dtype)) * temperature
        if cache is not None and cache.shape[2] != 0:
            cache_size = cache.shape[2]
            z[:, :, :cache_size] = cache[:, :, :, 0]
            mu[:, :, :cache_size] = cache[:, :, :, 1]
        z_cache = torch.concat([z[:, :, :prompt_len], z[:, :, -34:]], dim=2) if prompt_len > 0 else z[:, :, -34:]
        mu_cache = torch.concat([mu[:, :, :prompt_len], mu[:, :, -34:]], dim=2) if prompt_len > 0 else mu[:, :, -34:]
        new_cache = torch.stack([z_cache, mu_cache], dim=-1)
        t_span = torch.linspace(0, 1, n_timesteps + 1, device=mu.device, dtype=mu.dtype)
        if self.t_scheduler == 'cosine':                 # (flow_matching.py:67; any other value: the linear span)
            t_span = 1 - torch.cos(t_span * 0.5 * 3.14159265359)
        return self.solve_euler(z, t_span, mu, mask, spks, cond, num), new_cache

    @torch.no_grad()
    def solve_euler(self, x, t_span, mu, mask, spks, cond, num: Optional[Numerics] = None):
        """flow_model.py:100-135: the batch-of-2 trick -- row 0 conditional, row 1 unconditional (zero mu / spk /
        cond) -- one estimator forward (HIP kernels) per step, guidance (1+w) v_c - w v_u."""
        num = num or Numerics(dtype=x.dtype)
        t, dt = t_span[0].unsqueeze(0), t_span[1] - t_span[0]
        T = x.size(2)
        x_in = torch.zeros([2, 80, T], device=x.device, dtype=x.dtype)
        mask_in = torch.zeros([2, 1, T], device=x.device, dtype=x.dtype)
        mu_in = torch.zeros([2, 80, T], device=x.device, dtype=x.dtype)
        t_in = torch.zeros([2], device=x.device, dtype=x.dtype)
        spks_in = torch.zeros([2, 80], device=x.device, dtype=x.dtype)
        cond_in = torch.zeros([2, 80, T], device=x.device, dtype=x.dtype)
        for step in range(1, len(t_span)):
            x_in[:] = x
            mask_in[:] = mask
            mu_in[0] = mu
            t_in[:] = t.unsqueeze(0)
            spks_in[0] = spks
            cond_in[0] = cond
            d = self.estimator(x_in, mask_in, mu_in, t_in, spks_in, cond_in, dtype=num.dtype, gelu=num.gelu)
            d, d_u = torch.split(d, [x.size(0), x.size(0)], dim=0)
            x = x + dt * ((1.0 + self.inference_cfg_rate) * d - self.inference_cfg_rate * d_u)
            t = t + dt
            if step < len(t_span) - 1:
                dt = t_span[step + 1] - t
        return x.float()

    def compute_loss(self, x1, mask, mu, spks=None, cond=None, prompt_lens=None, draws=None, num: Optional[Numerics] = None):
        """Reference signature: x1,mu,cond (B,80,T) normalised mel; mask (B,1,T); spks (B,80)."""
        num = num or Numerics(dtype=mu.dtype)
        B, _, T = mu.shape
        length = mask.reshape(B, T).sum(dim=1).to(torch.int32)
        mu_cl = mu.transpose(1, 2).reshape(B * T, -1).to(num.dtype)
        cond_cl = None if cond is None else cond.transpose(1, 2).reshape(B * T, -1).to(num.dtype).contiguous()
        loss, xin = self.compute_loss_cl(x1.transpose(1, 2), mu_cl, spks.to(num.dtype), length, B, T, num, draws, cond=cond_cl,
                                         prompt_lens=prompt_lens)
        y = xin[:, :x1.shape[1]].reshape(B, T, -1).transpose(1, 2)
        return loss, y


class MaskedDiffWithXvec(nn.Module):
    def __init__(self, input_size: int = 512, output_size: int = 80, spk_embed_dim: int = 192, vocab_size: int = 4096,
                 input_frame_rate: int = 50, encoder: Optional[nn.Module] = None,
                 length_regulator: Optional[nn.Module] = None, decoder: Optional[nn.Module] = None):
        super().__init__()
        assert encoder is not None
        self.input_size, self.output_size, self.vocab_size = input_size, output_size, vocab_size
        self.input_frame_rate = input_frame_rate
        self.input_embedding = nn.Embedding(vocab_size, input_size)
        self.spk_embed_affine_layer = nn.Linear(spk_embed_dim, output_size)
        self.encoder = encoder
        self.encoder_proj = nn.Linear(self.encoder.output_size(), output_size)
        self.decoder = decoder
        self.length_regulator = length_regulator
        self.mel_mean, self.mel_std = MEL_MEAN, MEL_STD
        self.numerics = Numerics()

    def normalize_mel(self, mel):
        return (mel - self.mel_mean) / self.mel_std

    def denormalize_mel(self, mel):
        return mel * self.mel_std + self.mel_mean

    def _embedding_table(self, dtype):
        w = self.input_embedding.weight
        if dtype == w.dtype:
            return w.detach()
        from .modules import _cached
        return _cached(self.input_embedding, "tab", w, dtype, lambda: w.detach().to(dtype))

    def forward_no_prompt(self, batch: dict, device, draws=None) -> Dict[str, Any]:
        """llm_flow_model.py:181-229: no-prompt flow-matching loss (conditioning all zero)."""
        num = self.numerics
        dt = num.dtype
        token = batch['speech_token'].to(device)
        B, Lt = token.shape
        feat = batch['speech_feat'].to(device)
        T = feat.shape[1]
        tok_len = to_len(batch['speech_token_len'], device)
        feat_len = to_len(batch['speech_feat_len'], device)
        with torch.no_grad():
            spk = hip_linear(self.spk_embed_affine_layer, HF.l2norm_rows(batch['embedding'].to(device), dt))
            tok = HF.embed_gather(token, self._embedding_table(dt), tok_len)
        h = self.encoder.forward_cl(tok, B, Lt, tok_len, num, causal=False)
        h = hip_linear(self.encoder_proj, h)
        # `_true_dims` (train_joint.Trainer, shape-bucketed batches): device int32 [Lt_max, T_max, ceil(T_max/2), ...] of the
        # exact batch; absent = the tensors have their exact shapes
        td = batch.get('_true_dims')
        td = None if td is None else td.to(device)
        mu = self.length_regulator.forward_cl(h, B, Lt, T, feat_len, None if td is None else td[0:2])
        loss, _ = self.decoder.compute_loss_cl(feat, mu, spk, feat_len, B, T, num, draws, self.mel_mean, self.mel_std,
                                               t_true=None if td is None else td[1:])
        return {'loss': loss}

    def prompt_plan(self, feat_len, cross_len=None):
        """Host side of the anti-leakage strategies (flow_model.py:320-386), drawing from `random` in the reference's
        order: per utterance prompt dropout -> dynamic prompt length -> [cross-sample clamp] -> [silence band] -> text
        blinding.  Returns per-utterance dicts {total, copy, cross, silence, blind}."""
        C = ANTI_LEAKAGE_CONFIG
        plan = []
        for i, j in enumerate(feat_len):
            j = int(j)
            if C.get('prompt_dropout_enabled', True) and random.random() < C.get('prompt_dropout_prob', 0.10):
                plan.append(dict(total=0, copy=0, cross=False, silence=0, blind=False))
                continue
            if C.get('dynamic_prompt_enabled', True):
                lo = max(1, int(C.get('prompt_min_ratio', 0.10) * j))
                hi = max(lo + 1, int(C.get('prompt_max_ratio', 0.30) * j))
                pl = random.randint(lo, hi)
            else:
                pl = max(1, int(0.3 * j))
            cross = bool(C.get('cross_sample_enabled', True) and cross_len is not None and int(cross_len[i]) > 0)
            if cross:
                pl = min(pl, int(cross_len[i]))
            sil = 0
            if C.get('silence_padding_enabled', False):
                st = random.randint(C.get('silence_min_tokens', 5), C.get('silence_max_tokens', 10))
                frames = max(3, min(int(st * 22050 / 256 / self.input_frame_rate), 20))
                if pl + frames < j:
                    sil = frames
            blind = bool(C.get('text_blinding_enabled', True) and random.random() < C.get('text_blinding_prob', 0.7))
            plan.append(dict(total=pl + sil, copy=pl, cross=cross, silence=sil, blind=blind))
        return plan

    def forward_with_prompt(self, batch: dict, device, draws=None, plan=None) -> Dict[str, Any]:
        """flow_model.py:248-400: flow-only training with a mel prompt as conditioning and the
        anti-leakage strategies -- dynamic prompt length, prompt dropout, cross-sample prompts, optional silence band,
        text blinding of the encoder output, prompt-region loss mask + boundary weight, prompt-isolation attention."""
        num = self.numerics
        dt = num.dtype
        token = batch['speech_token'].to(device)
        B, Lt = token.shape
        feat = batch['speech_feat'].to(device).float()
        T = feat.shape[1]
        tok_len = to_len(batch['speech_token_len'], device)
        feat_len = to_len(batch['speech_feat_len'], device)
        with torch.no_grad():
            spk = hip_linear(self.spk_embed_affine_layer, HF.l2norm_rows(batch['embedding'].to(device), dt))
            tok = HF.embed_gather(token, self._embedding_table(dt), tok_len)
        h = self.encoder.forward_cl(tok, B, Lt, tok_len, num, causal=False)
        h = hip_linear(self.encoder_proj, h)
        mu = self.length_regulator.forward_cl(h, B, Lt, T, feat_len)
        cross = batch.get('cross_sample_mel')
        cross_len = batch.get('cross_sample_mel_len') if cross is not None else None
        if plan is None:
            plan = self.prompt_plan(batch['speech_feat_len'].tolist(), None if cross_len is None else cross_len.tolist())
        featn = self.normalize_mel(feat)
        crossn = None if cross is None else self.normalize_mel(cross.to(device).float())
        sil_val = (ANTI_LEAKAGE_CONFIG.get('silence_mel_value', -11.5) - self.mel_mean) / self.mel_std
        conds = torch.zeros_like(featn)
        keep_rows = torch.ones(B, T, device=device, dtype=dt)
        for i, pl in enumerate(plan):
            c = pl['copy']
            if c > 0:
                conds[i, :c] = crossn[i, :c] if pl['cross'] else featn[i, :c]
                if pl['silence'] > 0:
                    conds[i, c:c + pl['silence']] = sil_val
                if pl['blind']:
                    keep_rows[i, :c] = 0.0            # text blinding: the encoder output under the prompt is zeroed
        mu = mu * keep_rows.reshape(B * T, 1)
        loss, _ = self.decoder.compute_loss_cl(feat, mu, spk, feat_len, B, T, num, draws, self.mel_mean, self.mel_std,
                                               cond=conds.reshape(B * T, -1).to(dt).contiguous(),
                                               prompt_lens=[pl['total'] for pl in plan])
        return {'loss': loss}

    def forward(self, batch: dict, device) -> Dict[str, Any]:
        """flow_model.py:248-318: the no-prompt mode when NO_PROMPT_TRAINING_CONFIG['enabled'] ('full': conditioning all
        zero; 'mixed' :437-455: per utterance, with probability 1 - no_prompt_ratio, a short own-mel prompt with its loss
        mask / isolation and none of the anti-leakage strategies), else the with-prompt path."""
        if NO_PROMPT_TRAINING_CONFIG.get('enabled', False):
            if NO_PROMPT_TRAINING_CONFIG.get('mode', 'full') == 'full':
                return self.forward_no_prompt(batch, device)
            ratio = NO_PROMPT_TRAINING_CONFIG.get('no_prompt_ratio', 0.8)
            plan = []
            for j in batch['speech_feat_len'].tolist():
                pl = 0 if random.random() < ratio else random.randint(1, max(2, int(0.1 * int(j))))
                plan.append(dict(total=pl, copy=pl, cross=False, silence=0, blind=False))
            return self.forward_with_prompt(batch, device, plan=plan)
        return self.forward_with_prompt(batch, device)

    @staticmethod
    def ode_steps_for(n_frames: int) -> int:
        """flow_model.py:525-536: Euler steps by sequence length."""
        return 20 if n_frames > 500 else 15 if n_frames > 300 else 10

    def _encode_one(self, token, embedding, device):
        """spk projection + token embedding + encoder + encoder_proj for one utterance: ([L, 80], [1, 80])."""
        num, dt = self.numerics, self.numerics.dtype
        L = token.shape[1]
        ln = torch.tensor([L], dtype=torch.int32, device=device)
        spk = hip_linear(self.spk_embed_affine_layer, HF.l2norm_rows(embedding.to(device), dt))
        tok = HF.embed_gather(token.to(device), self._embedding_table(dt), ln)
        h = self.encoder.forward_cl(tok, 1, L, ln, num, causal=False)
        return hip_linear(self.encoder_proj, h), spk

    @torch.no_grad()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding,
                  flow_cache=None, noise: Optional[torch.Tensor] = None):
        """flow_model.py:474-551 (batch 1): prompt + target tokens through the encoder, head/mid/tail length regulation,
        the prompt mel as conditioning, Euler steps by length (HIP estimator); returns (target mel (1,80,mel_len2) fp32,
        new cache).  Like the reference it does not normalise the mel.  `noise` pins the sampler's initial z."""
        assert token.shape[0] == 1
        device = self.input_embedding.weight.device
        n1, n2 = prompt_token.shape[1], token.shape[1]
        h, spk = self._encode_one(torch.cat([prompt_token, token], dim=1), embedding, device)
        mel1, mel2 = prompt_feat.shape[1], int(n2 / self.input_frame_rate * 22050 / 256)
        mu = self.length_regulator.inference_cl(h[:n1], h[n1:], mel1, mel2, self.input_frame_rate)
        T = mel1 + mel2
        cond = torch.zeros(1, T, self.output_size, device=device, dtype=torch.float32)
        cond[:, :mel1] = prompt_feat.to(device)
        feat, cache = self.decoder(mu=mu.float().t().unsqueeze(0).contiguous(), mask=torch.ones(1, 1, T, device=device),
                                   n_timesteps=self.ode_steps_for(T), spks=spk.float(), cond=cond.transpose(1, 2).contiguous(),
                                   prompt_len=mel1, cache=flow_cache, noise=noise, num=self.numerics)
        return feat[:, :, mel1:].float(), cache

    @torch.no_grad()
    def inference_like_training(self, token, token_len, feat_len, embedding, prompt_feat=None, prompt_len=0, n_timesteps=10,
                                noise: Optional[torch.Tensor] = None):
        """flow_model.py:553-638 (batch 1): the whole token sequence regulated to feat_len frames as in training, optional
        prompt conditioning of prompt_len frames -> full mel (1,80,feat_len) fp32."""
        assert token.shape[0] == 1
        device = self.input_embedding.weight.device
        T = int(feat_len.item()) if torch.is_tensor(feat_len) else int(feat_len)
        h, spk = self._encode_one(token, embedding, device)
        mu = self.length_regulator.forward_cl(h, 1, token.shape[1], T, torch.tensor([T], dtype=torch.int32, device=device))
        cond = torch.zeros(1, T, self.output_size, device=device, dtype=torch.float32)
        if prompt_feat is not None and prompt_len > 0:
            n = min(prompt_len, prompt_feat.shape[1], T)
            cond[:, :n] = prompt_feat[:, :n].to(device)
        if n_timesteps is None or n_timesteps == 10:
            n_timesteps = self.ode_steps_for(T)
        feat, _ = self.decoder(mu=mu.float().t().unsqueeze(0).contiguous(), mask=torch.ones(1, 1, T, device=device),
                               n_timesteps=n_timesteps, spks=spk.float(), cond=cond.transpose(1, 2).contiguous(),
                               prompt_len=prompt_len if prompt_feat is not None else 0, cache=None, noise=noise, num=self.numerics)
        return feat.float()


def build_flow_model(pretrained_path: Optional[str] = None, device: str = 'cuda', input_size: int = 512,
                     output_size: int = 80, spk_embed_dim: int = 192, vocab_size: int = 4096,
                     encoder_attention_heads: int = 8, encoder_linear_units: int = 2048, encoder_num_blocks: int = 6,
                     decoder_channels: tuple = (256, 256), decoder_attention_head_dim: int = 64,
                     decoder_n_blocks: int = 4, decoder_num_mid_blocks: int = 12, decoder_num_heads: int = 8,
                     numerics: Optional[Numerics] = None) -> MaskedDiffWithXvec:
    """reference flow_model.py:641-767 (CosyVoice-300M dims by default)."""
    numerics = numerics or Numerics()
    encoder = RelPosEncoder(input_size=input_size, output_size=input_size, attention_heads=encoder_attention_heads,
                            linear_units=encoder_linear_units, num_blocks=encoder_num_blocks, dropout_rate=0.1,
                            positional_dropout_rate=0.1, attention_dropout_rate=0.1, input_layer="linear",
                            kind="conformer", static_chunk_size=0, ln_eps=numerics.enc_ln_eps)
    regulator = InterpolateRegulator(channels=output_size, sampling_ratios=(1, 1, 1, 1), out_channels=output_size, groups=1)
    estimator = ConditionalDecoder(in_channels=320, out_channels=80, channels=decoder_channels, dropout=0.0,
                                   attention_head_dim=decoder_attention_head_dim, n_blocks=decoder_n_blocks,
                                   num_mid_blocks=decoder_num_mid_blocks, num_heads=decoder_num_heads, act_fn='gelu')
    decoder = ConditionalCFM(in_channels=output_size, n_spks=1, spk_emb_dim=output_size, sigma_min=1e-6,
                             t_scheduler='cosine', training_cfg_rate=0.2, inference_cfg_rate=0.7, estimator=estimator)
    model = MaskedDiffWithXvec(input_size=input_size, output_size=output_size, spk_embed_dim=spk_embed_dim,
                               vocab_size=vocab_size, input_frame_rate=50, encoder=encoder,
                               length_regulator=regulator, decoder=decoder)
    model.numerics = numerics
    if pretrained_path is not None:
        wf = os.path.join(pretrained_path, 'flow.pt') if os.path.isdir(pretrained_path) else pretrained_path
        if os.path.exists(wf):
            sd = torch.load(wf, map_location='cpu', weights_only=True)
            try:
                model.load_state_dict(sd, strict=True)
                print("Weights loaded successfully (strict=True)")
            except Exception as e:   # tolerant loading, as the reference does
                print(f"Strict loading failed: {e}\nAttempting partial loading...")
                own = model.state_dict()
                ok = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
                own.update(ok)
                model.load_state_dict(own, strict=False)
                print(f"Partial loading: {len(ok)}/{len(sd)} weights loaded")
        else:
            print(f"Warning: Weight file not found: {wf}\nUsing random initialization")
    return model.to(device)
