"""Length / mask helpers with the reference's names (utils.py:20-109 == cosyvoice/utils/mask.py).

The HIP path never materialises these masks (kernels consume int32 lengths); the functions
exist for API compatibility and run on whatever device `lengths` lives on.  `make_pad_mask`
with max_len=0 needs `lengths.max()`; pass `max_len` to stay sync-free."""
import random

import numpy as np
import torch

IGNORE_ID = -1


def set_all_random_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """True at padded positions.  make_pad_mask([5,3,2]) -> [[0,0,0,0,0],[0,0,0,1,1],[0,0,1,1,1]]."""
    n = max_len if max_len > 0 else int(lengths.max().item())
    return torch.arange(n, dtype=torch.int64, device=lengths.device)[None, :] >= lengths[:, None]


def subsequent_chunk_mask(size: int, chunk_size: int, num_left_chunks: int = -1,
                          device: torch.device = torch.device("cpu")) -> torch.Tensor:
    pos = torch.arange(size, device=device)
    limit = (torch.div(pos, chunk_size, rounding_mode='trunc') + 1) * chunk_size
    return pos[None, :] < limit[:, None]


def add_optional_chunk_mask(xs, masks, use_dynamic_chunk, use_dynamic_left_chunk, decoding_chunk_size,
                            static_chunk_size, num_decoding_left_chunks, enable_full_context=True):
    if use_dynamic_chunk:
        raise NotImplementedError("dynamic chunk training is not used by CosyVoice-300M")
    if static_chunk_size > 0:
        cm = masks & subsequent_chunk_mask(xs.size(1), static_chunk_size, num_decoding_left_chunks, xs.device)[None]
    else:
        cm = masks
    assert cm.dtype == torch.bool
    dead = cm.sum(dim=-1) == 0
    if bool(dead.any()):
        cm = cm.clone()
        cm[dead] = True
    return cm


def mask_to_bias(mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    assert mask.dtype == torch.bool
    assert dtype in (torch.float32, torch.bfloat16, torch.float16)
    return (1.0 - mask.to(dtype)) * -1.0e+10


def pad_list(xs, pad_value: int):
    n = max(len(x) for x in xs)
    out = xs[0].new_full((len(xs), n) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, :len(x)] = x
    return out
