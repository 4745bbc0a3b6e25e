"""Shared builders for the model-level tests (product models from the golden metadata)."""
import torch

from oracle.detweights import det_state_dict


def tup(d):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}


def build_flow_product(meta_flow, device, numerics, lora=True, seed=None):
    from cosyvoice_lora_finetune_framework_amd.flow_model import build_flow_model
    from cosyvoice_lora_finetune_framework_amd.lora import apply_lora_to_model
    m = build_flow_model(None, 'cpu', numerics=numerics, **tup(meta_flow.get("build", {})))
    if lora:
        L = meta_flow["lora"]
        apply_lora_to_model(m, r=L["r"], lora_alpha=L["alpha"], lora_dropout=0.0, target_modules=L["targets"])
    spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    if lora:
        m.load_state_dict(det_state_dict(spec, meta_flow["weight_seed"] if seed is None else seed), strict=True)
    return m.to(device).eval()


def build_llm_product(meta_llm, device, numerics, lora=True, full=False):
    from cosyvoice_lora_finetune_framework_amd.llm_model import build_llm_model
    from cosyvoice_lora_finetune_framework_amd.lora import apply_lora_to_model
    if full:
        m = build_llm_model(None, 'cpu', numerics=numerics)
    else:
        c = meta_llm["build"]
        m = build_llm_model(None, 'cpu', numerics=numerics, text_encoder_input_size=c['text_in'], llm_input_size=c['d'],
                            llm_output_size=c['d'], text_token_size=c['text_vocab'], speech_token_size=c['speech_vocab'],
                            attention_heads=c['heads'], linear_units=c['ff'], text_encoder_blocks=c['text_blocks'],
                            llm_blocks=c['llm_blocks'])
    if lora:
        L = meta_llm["lora"]
        apply_lora_to_model(m, r=L["r"], lora_alpha=L["alpha"], lora_dropout=0.0, target_modules=L["targets"])
        spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        m.load_state_dict(det_state_dict(spec, meta_llm["weight_seed"]), strict=True)
    return m.to(device).eval()


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def lora_grads(model):
    return {n: p.grad for n, p in model.named_parameters() if 'lora_' in n and p.grad is not None}


def keep_fields_host(seed: int, site: int, ngroups: int):
    """Host replica of csrc/common.h cvft_drop_key / cvft_keep4: the 16-bit fields of the mask draws of groups 0 .. ngroups-1 (group g
    = elements 4g .. 4g+3 of the flat tensor) as a numpy uint32 array [ngroups, 4]; element e is KEPT when its field >= thr,
    thr = min(65535, rint(p * 65536)).  Every device kernel that draws a mask (cvft_dropout_add, the dropout-skinny product, the
    masked rank extension of cvft_gemm, GEMM output dropout, the q|k|v chain kernels) calls that one function."""
    import numpy as np
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(z):
        z = (z + np.uint64(0x9e3779b97f4a7c15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & M
        return z ^ (z >> np.uint64(31))

    def fmix32(h):
        h = h ^ (h >> np.uint32(16))
        h = (h.astype(np.uint64) * np.uint64(0x85ebca6b)).astype(np.uint32)
        h = h ^ (h >> np.uint32(13))
        h = (h.astype(np.uint64) * np.uint64(0xc2b2ae35)).astype(np.uint32)
        return h ^ (h >> np.uint32(16))
    with np.errstate(over="ignore"):
        key = int(mix(np.uint64(seed) ^ (np.uint64(site) << np.uint64(32))))
        mul = np.uint64(((key >> 17) & 0xFFFFFFFF) | 1)
        g = (np.arange(ngroups, dtype=np.uint64) * mul).astype(np.uint32)          # per-site odd multiplier (mod 2^32)
        lo = fmix32(g ^ np.uint32(key & 0xFFFFFFFF))
        hi = fmix32(g ^ np.uint32(key >> 32))
    f = np.uint32(0xFFFF)
    return np.stack([lo & f, lo >> np.uint32(16), hi & f, hi >> np.uint32(16)], 1)


def drop_thr_host(p: float) -> int:
    import numpy as np
    return int(min(65535.0, float(np.rint(np.float32(p) * np.float32(65536.0)))))
