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
