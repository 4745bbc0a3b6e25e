"""smoke: one tiny joint LLM+Flow LoRA training step of the HIP hot path on `device`, checked
against the CPU oracle (oracle/ is imported here ONLY as the checker)."""
import json
import os

import torch


def run(device) -> None:
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import ref_math as R
    from oracle.detweights import det_state_dict
    from .flow_model import build_flow_model
    from .llm_flow_model import JointLLMFlowModel
    from .llm_model import build_llm_model
    from .lora import apply_lora_to_model
    from .modules import Numerics
    from .optim import FlatAdamW
    from .synthetic import cfm_draws, synth_batch
    from .hipops import binding as cb

    assert os.path.exists(cb.LIB_PATH), "libcvft.so missing: the HIP extension must be built (no fallback)"
    num = Numerics(dtype=torch.float32)
    flow = build_flow_model(None, 'cpu', numerics=num, input_size=128, vocab_size=64, encoder_attention_heads=2,
                            encoder_linear_units=256, encoder_num_blocks=2, decoder_channels=(64, 64),
                            decoder_n_blocks=1, decoder_num_mid_blocks=2, decoder_num_heads=2)
    llm = build_llm_model(None, 'cpu', numerics=num, text_encoder_input_size=64, llm_input_size=128, llm_output_size=128,
                          text_token_size=100, speech_token_size=50, attention_heads=2, linear_units=256,
                          text_encoder_blocks=2, llm_blocks=2)
    apply_lora_to_model(flow, r=4, lora_alpha=8, lora_dropout=0.0,
                        target_modules=['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'w_1', 'w_2'])
    apply_lora_to_model(llm, r=4, lora_alpha=8, lora_dropout=0.0,
                        target_modules=['linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2'])
    for m, seed in ((flow, 3), (llm, 5)):
        m.load_state_dict(det_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], seed))
    sd_flow = {k: v.clone() for k, v in flow.state_dict().items()}
    sd_llm = {k: v.clone() for k, v in llm.state_dict().items()}
    jm = JointLLMFlowModel(llm, flow, 'joint', llm_loss_weight=2.0, flow_loss_weight=1.0).to(device).eval()   # dropout off: the oracle has none
    batch = synth_batch([24, 17], text_lens=[7, 5], token_lens=[13, 9], seed=11, text_vocab=100, speech_vocab=50)
    draws = cfm_draws(2, 24, seed=77)
    opt = FlatAdamW([p for p in jm.parameters() if p.requires_grad], lr=1e-3)
    out = jm(batch, device, draws)
    out['loss'].backward()
    opt.step()
    torch.cuda.synchronize()
    cfg = R.OracleConfig(flow_lora_scale=2.0, llm_lora_scale=2.0, speech_token_size=50)
    ref = R.joint_forward(sd_llm, sd_flow, batch, draws, cfg, 'joint', 2.0, 1.0)
    err = abs(float(out['loss']) - float(ref['loss'])) / abs(float(ref['loss']))
    print(json.dumps({"smoke": "joint tiny step", "hip_loss": float(out['loss']), "oracle_loss": float(ref['loss']),
                      "rel_err": err, "grad_norm": float(opt.grad_norm())}))
    assert err < 1e-4, err
    assert torch.isfinite(opt.flat_p).all()
