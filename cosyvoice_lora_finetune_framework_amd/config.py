"""Configuration dicts -- same names and keys as the reference's config.py (config.py:60-282),
so code written against ``from config import JOINT_TRAINING_CONFIG`` keeps working.  Values
are the reference's shipped defaults (8 GB-laptop settings); MI355X-specific knobs live in
``MI355X_CONFIG`` (new) and override nothing unless a caller asks for them."""
import os

PROJECT_ROOT = os.path.dirname(os.path.abspath(__file__))
PRETRAINED_MODEL_DIR = os.environ.get(
    "COSYVOICE_PRETRAINED_DIR", os.path.join(PROJECT_ROOT, "pretrained_models", "CosyVoice-300M"))
DATA_DIR = os.environ.get("COSYVOICE_DATA_DIR", os.path.join(PROJECT_ROOT, "data"))
RAW_AUDIO_DIR = os.path.join(PROJECT_ROOT, "raw_audio")
OUTPUT_DIR = os.environ.get("COSYVOICE_OUTPUT_DIR", os.path.join(PROJECT_ROOT, "output"))

TRAIN_CONFIG = {                      # config.py:60-81
    'max_epochs': 100, 'batch_size': 2, 'accumulate_grad_batches': 4, 'learning_rate': 1e-4,
    'min_learning_rate': 1e-6, 'weight_decay': 0.01, 'warmup_steps': 50, 'max_feat_len': 600,
    'precision': '16-mixed', 'gradient_clip_val': 1.0, 'augmentation': True,
}

LORA_CONFIG = {                       # config.py:84-100
    'use_lora': True, 'lora_r': 16, 'lora_alpha': 16, 'lora_dropout': 0.05,
    'target_modules': ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2'],
}

ANTI_LEAKAGE_CONFIG = {               # config.py:108-145
    'silence_padding_enabled': False, 'silence_token_id': 0, 'silence_min_tokens': 5, 'silence_max_tokens': 10,
    'silence_mel_value': -11.5, 'dynamic_prompt_enabled': True, 'prompt_min_ratio': 0.05, 'prompt_max_ratio': 0.20,
    'prompt_dropout_enabled': True, 'prompt_dropout_prob': 0.25, 'boundary_loss_enabled': True, 'boundary_frames': 25,
    'boundary_loss_weight': 5.0, 'cross_sample_enabled': True, 'cross_sample_prob': 0.85, 'text_blinding_enabled': True,
    'text_blinding_prob': 0.95, 'text_blinding_mode': 'zero',
}

NO_PROMPT_TRAINING_CONFIG = {'enabled': False, 'mode': 'full', 'no_prompt_ratio': 0.8, 'use_mean_embedding': False}

JOINT_TRAINING_CONFIG = {             # config.py:179-224
    'training_mode': 'joint', 'llm_loss_weight': 2.0, 'flow_loss_weight': 1.0, 'no_prompt_training': True,
    'llm_lora': {'lora_r': 8, 'lora_alpha': 16, 'lora_dropout': 0.15,
                 'target_modules': ['linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2']},
    'flow_lora': {'lora_r': 16, 'lora_alpha': 32, 'lora_dropout': 0.05,
                  'target_modules': ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'w_1', 'w_2']},
    'learning_rate': 2e-4, 'max_epochs': 100, 'batch_size': 1, 'accumulate_grad_batches': 16, 'max_feat_len': 250,
}

MEL_MEAN = -6.0                       # config.py:241
MEL_STD = 2.0                         # config.py:242

INFERENCE_CONFIG = {
    'max_prompt_seconds': 5, 'physical_trim_enabled': True, 'physical_trim_mode': 'absolute',
    'physical_trim_frames': 80, 'physical_trim_extra_ms': 300, 'trim_ratio': 0.08, 'boundary_trim_ratio': 0.20,
}

MODEL_CONFIG = {'input_size': 512, 'output_size': 80, 'spk_embed_dim': 192, 'vocab_size': 4096,
                'input_frame_rate': 50, 'sample_rate': 22050}

# CosyVoice-300M LLM dims: not pinned in the reference tree (they live in the un-shipped
# pretrained cosyvoice.yaml); public upstream values (SURVEY.md section 8 preamble).
LLM_MODEL_CONFIG = {
    'text_encoder_input_size': 512, 'llm_input_size': 1024, 'llm_output_size': 1024, 'text_token_size': 51866,
    'speech_token_size': 4096, 'spk_embed_dim': 192,
    'text_encoder': {'output_size': 1024, 'attention_heads': 16, 'linear_units': 4096, 'num_blocks': 6},
    'llm': {'output_size': 1024, 'attention_heads': 16, 'linear_units': 4096, 'num_blocks': 14},
}

# New (no reference counterpart): MI355X run settings for the BASELINE.json configs.
MI355X_CONFIG = {
    'compute_dtype': 'bf16',          # 'fp32' = exact-fp32 MFMA parity path
    'max_feat_len': 1000,             # BASELINE shapes are 500 / 1000 frames; the reference's 250 would truncate
    'flow_lora_r': 16, 'llm_lora_r': 16, 'lora_alpha_over_r': 2.0,
    'per_gpu_batch': 16,
}
