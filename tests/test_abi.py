"""The C-ABI library loads (no GPU needed) and exports every symbol include/cvft.h declares;
the ctypes table in hipops/binding.py covers exactly the same set."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "cvft.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cvft_[a-z0-9_]+)\s*\(", src)) - {"cvft_gemm_args"})


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("cvft_gemm", "cvft_tn_accum", "cvft_attn_bias_fwd", "cvft_attn_bias_bwd", "cvft_attn_relpos_fwd",
                 "cvft_attn_relpos_bwd", "cvft_layernorm_fwd", "cvft_layernorm_bwd", "cvft_groupnorm_mish_fwd",
                 "cvft_groupnorm_mish_bwd", "cvft_dwconv1d_fwd", "cvft_dwconv1d_bwd", "cvft_time_embed",
                 "cvft_cfm_prepare", "cvft_masked_mse_fwd", "cvft_masked_mse_bwd", "cvft_ce_fwd", "cvft_ce_bwd",
                 "cvft_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (bind to torch's HIP runtime first)
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    if not os.path.exists(cb.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = cb.lib()
    declared = _declared()
    assert sorted(cb.SIGNATURES.keys()) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cvft_version() >= 100


def test_ops_fail_loudly_without_gpu_tensors():
    import torch
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    with pytest.raises(cb.CvftError):
        HF.gemm(torch.zeros(4, 8), torch.zeros(4, 8))
