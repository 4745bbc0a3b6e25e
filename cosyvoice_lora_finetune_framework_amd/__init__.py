"""MI355X-native joint LLM+Flow LoRA fine-tuning hot path for CosyVoice-300M.

Drop-in for the reference's ``lora.py`` / ``llm_flow_model.py`` / ``train_joint.py`` API
(SURVEY.md section 8b); compute runs in hand-written gfx950 HIP kernels behind the
C-ABI library ``libcvft.so`` (include/cvft.h).  Sub-modules are imported lazily so that
host-only helpers (``synthetic``, ``config``) work without a GPU.
"""
__version__ = "0.1.0"
