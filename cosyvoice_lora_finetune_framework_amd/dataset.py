"""On-disk data path of the joint fine-tuning step: parquet shards -> samples -> padded batch dict.

Same public surface and batch contract as the reference's ``dataset.py`` (``MelAugmentation`` 28-160,
``FlowFinetuneDataset`` 168-482, ``collate_fn`` 485-596, ``create_dataloader`` 599-621) so that
``train_joint.py`` can be pointed at real shards written by ``prepare_joint_data.py`` (columns ``utt, text, text_token,
speech_token, speech_feat`` (flattened) ``, speech_feat_shape, utt_embedding, spk_embedding``; 275-284, 365-372).

Host-side only (the hot path starts at the batch dict).  What is deliberately identical to the reference, because
seeded runs must reproduce its batches: the order of ``random`` / ``torch`` draws inside the augmentation and the
cross-sample prompt choice, the proportional truncation rule of the collate step, the pad values (token 0, mel
-11.5) and the rule that ``text_token`` is emitted only when every sample of the batch has one.
New here: ``ShardSampler`` -- the data-parallel partition of SURVEY 8e (same shuffled index list on every rank,
rank-strided slices, whole global batches only)."""
from __future__ import annotations

import math
import os
import random
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset, Sampler

try:
    from .config import ANTI_LEAKAGE_CONFIG, JOINT_TRAINING_CONFIG
except Exception:                                        # pragma: no cover - config is part of the package
    ANTI_LEAKAGE_CONFIG = {'cross_sample_enabled': True, 'cross_sample_prob': 0.5}
    JOINT_TRAINING_CONFIG = {'max_feat_len': 150}

MEL_PADDING_VALUE = -11.5          # log-mel of silence (dataset.py:526)
N_MELS = 80


# ---------------------------------------------------------------------------------
# augmentation (dataset.py:28-160)
# ---------------------------------------------------------------------------------
class MelAugmentation:
    """SpecAugment-style time / frequency masks, log-gain, +-5 % time stretch (tokens resampled alongside), Gaussian
    noise -- each gated by one ``random.random()`` draw, in this order."""

    def __init__(self, enable: bool = True, time_mask_prob: float = 0.5, time_mask_max_ratio: float = 0.1,
                 num_time_masks: int = 2, freq_mask_prob: float = 0.5, freq_mask_max_bins: int = 8, num_freq_masks: int = 2,
                 volume_prob: float = 0.5, volume_range: tuple = (-0.2, 0.2), time_stretch_prob: float = 0.3,
                 time_stretch_range: tuple = (0.95, 1.05), noise_prob: float = 0.3, noise_std: float = 0.02):
        self.enable = enable
        self.time_mask_prob, self.time_mask_max_ratio, self.num_time_masks = time_mask_prob, time_mask_max_ratio, num_time_masks
        self.freq_mask_prob, self.freq_mask_max_bins, self.num_freq_masks = freq_mask_prob, freq_mask_max_bins, num_freq_masks
        self.volume_prob, self.volume_range = volume_prob, volume_range
        self.time_stretch_prob, self.time_stretch_range = time_stretch_prob, time_stretch_range
        self.noise_prob, self.noise_std = noise_prob, noise_std

    def __call__(self, mel: torch.Tensor, speech_token: Optional[torch.Tensor] = None):
        if not self.enable:
            return mel, speech_token
        mel = mel.clone()
        if random.random() < self.time_mask_prob:
            mel = self._time_mask(mel)
        if random.random() < self.freq_mask_prob:
            mel = self._freq_mask(mel)
        if random.random() < self.volume_prob:
            mel = self._volume_perturb(mel)
        if random.random() < self.time_stretch_prob and speech_token is not None:
            mel, speech_token = self._time_stretch(mel, speech_token)
        if random.random() < self.noise_prob:
            mel = self._add_noise(mel)
        return mel, speech_token

    def _time_mask(self, mel):
        T = mel.shape[0]
        for _ in range(self.num_time_masks):
            width = int(T * self.time_mask_max_ratio * random.random())
            if width > 0:
                start = random.randint(0, max(0, T - width))
                mel[start:start + width, :] = mel.mean()           # fill with the (current) global mean
        return mel

    def _freq_mask(self, mel):
        n_mels = mel.shape[1]
        for _ in range(self.num_freq_masks):
            width = random.randint(1, self.freq_mask_max_bins)
            start = random.randint(0, max(0, n_mels - width))
            mel[:, start:start + width] = mel.mean()
        return mel

    def _volume_perturb(self, mel):
        return mel + random.uniform(*self.volume_range)             # additive in the log domain

    def _time_stretch(self, mel, speech_token):
        T = mel.shape[0]
        factor = random.uniform(*self.time_stretch_range)
        new_T = int(T * factor)
        if new_T < 10 or new_T > T * 2:
            return mel, speech_token
        stretched = F.interpolate(mel.t().unsqueeze(0), size=new_T, mode='linear', align_corners=False).squeeze(0).t()
        n_tok = speech_token.shape[0]
        new_tok = int(n_tok * factor)
        if new_tok > 0:
            idx = torch.linspace(0, n_tok - 1, new_tok).long().clamp(0, n_tok - 1)
            speech_token = speech_token[idx]
        return stretched, speech_token

    def _add_noise(self, mel):
        return mel + torch.randn_like(mel) * self.noise_std


# ---------------------------------------------------------------------------------
# tolerant column decoding (dataset.py:342-482)
# ---------------------------------------------------------------------------------
def _as_tensor(value: Any, dtype: torch.dtype) -> torch.Tensor:
    """Parquet readers hand back tensors, numpy arrays, (nested) lists or arrays of arrays depending on the writer."""
    if isinstance(value, torch.Tensor):
        return value.to(dtype)
    if isinstance(value, np.ndarray):
        if value.dtype == object:                        # array of per-frame arrays
            value = np.stack([np.asarray(v) for v in value])
        return torch.from_numpy(np.array(value, copy=True)).to(dtype)
    if isinstance(value, list):
        return torch.tensor(value, dtype=dtype)
    return torch.tensor(np.array(value), dtype=dtype)


def decode_mel(value: Any, shape: Any = None, n_mels: int = N_MELS) -> Optional[torch.Tensor]:
    """-> (T, n_mels) fp32, or None when the stored data cannot be a mel of `n_mels` bins.  Flattened features are
    reshaped with the stored `speech_feat_shape` when it is a pair, else by `n_mels`; a (n_mels, T) matrix is transposed."""
    mel = _as_tensor(value, torch.float32)
    if mel.dim() == 1:
        if shape is not None and isinstance(shape, (list, tuple, np.ndarray)) and len(shape) == 2:
            mel = mel.view(int(shape[0]), int(shape[1]))
        elif mel.numel() % n_mels == 0:
            mel = mel.view(-1, n_mels)
        else:
            return None
    if mel.dim() != 2:
        return None
    if mel.shape[-1] != n_mels and mel.shape[0] == n_mels:
        mel = mel.transpose(0, 1)
    return mel


def resolve_shards(data_dir: str, verbose: bool = True) -> List[str]:
    """`data.list` entries may be absolute paths from the machine that wrote them (either slash style): try the path as
    written, its basename, and progressively shorter suffixes under `data_dir`; without a list, walk for *.parquet."""
    listing = os.path.join(data_dir, 'data.list')
    found: List[str] = []
    if os.path.exists(listing):
        with open(listing, 'r', encoding='utf-8') as f:
            entries = [ln.strip() for ln in f if ln.strip()]
        for raw in entries:
            raw = raw.replace('\\', '/')
            parts = raw.split('/')
            candidates = [raw, os.path.join(data_dir, os.path.basename(raw)), os.path.join(data_dir, raw)]
            if len(parts) > 1:
                candidates += [os.path.join(data_dir, parts[-1]), os.path.join(data_dir, '/'.join(parts[1:]))]
                if len(parts) > 2:
                    candidates.append(os.path.join(data_dir, '/'.join(parts[2:])))
            hit = next((c for c in candidates if os.path.exists(c)), None)
            if hit is not None:
                found.append(hit)
            elif verbose:
                print(f"Warning: Could not find parquet file for: {raw}")
    else:
        for root, _, files in os.walk(data_dir):
            found += [os.path.join(root, f) for f in files if f.endswith('.parquet')]
        found = sorted(found)
    return found


def read_shard(path: str) -> List[Dict[str, Any]]:
    """One parquet shard -> list of row dicts (pyarrow; list columns arrive as numpy arrays like pandas' records)."""
    import pyarrow.parquet as pq
    table = pq.read_table(path)
    cols = {name: table.column(name).to_pylist() for name in table.column_names}
    n = table.num_rows
    rows = []
    for i in range(n):
        row = {}
        for name, col in cols.items():
            v = col[i]
            row[name] = np.asarray(v) if isinstance(v, list) else v
        rows.append(row)
    return rows


class FlowFinetuneDataset(Dataset):
    """Samples of all shards under `data_dir`, decoded lazily in `__getitem__`; undecodable rows yield None (dropped by
    `collate_fn`).  With probability `cross_sample_prob` a sample carries another utterance's first <= 100 mel frames as
    `cross_sample_mel` (anti-leakage prompting, flow-only mode)."""

    def __init__(self, data_dir: str, max_duration: float = 15.0, target_sr: int = 22050, augmentation: bool = True,
                 verbose: bool = True):
        self.data_dir, self.max_duration, self.target_sr = data_dir, max_duration, target_sr
        self.hop_size, self.n_mels = 256, N_MELS
        self.augmentation = MelAugmentation(enable=augmentation)
        self.augmentation_enabled = augmentation
        self.samples: List[Dict[str, Any]] = []
        for shard in resolve_shards(data_dir, verbose):
            try:
                self.samples.extend(read_shard(shard))
            except Exception as e:                               # a corrupt shard must not end the run
                print(f"Failed to read {shard}: {e}")
        self.cross_sample_enabled = ANTI_LEAKAGE_CONFIG.get('cross_sample_enabled', True)
        self.cross_sample_prob = ANTI_LEAKAGE_CONFIG.get('cross_sample_prob', 0.5)
        if verbose:
            print(f"Dataset loaded: {len(self.samples)} samples (augmentation {'on' if augmentation else 'off'}, "
                  f"cross-sample prompting {'p=%.2f' % self.cross_sample_prob if self.cross_sample_enabled else 'off'})")

    def __len__(self) -> int:
        return len(self.samples)

    def _get_random_prompt_mel(self, exclude_idx: int, max_len: int = 100) -> Optional[torch.Tensor]:
        if len(self.samples) < 2:
            return None
        pick, attempts = exclude_idx, 0
        while attempts < 10 and pick == exclude_idx:
            pick = random.randint(0, len(self.samples) - 1)
            attempts += 1
        if pick == exclude_idx:
            return None
        other = self.samples[pick]
        if 'speech_feat' not in other:
            return None
        try:
            mel = decode_mel(other['speech_feat'], other.get('speech_feat_shape', None), self.n_mels)
        except Exception:
            return None
        return None if mel is None else mel[:max_len]

    def __getitem__(self, idx: int) -> Optional[Dict[str, Any]]:
        sample = self.samples[idx]
        try:
            if 'speech_feat' not in sample or 'speech_token' not in sample:
                return None
            mel = decode_mel(sample['speech_feat'], sample.get('speech_feat_shape', None), self.n_mels)
            if mel is None:
                return None
            token = _as_tensor(sample['speech_token'], torch.long).flatten()
            embedding = next((sample[k] for k in ('utt_embedding', 'spk_embedding', 'embedding')
                              if k in sample and sample[k] is not None), None)
            embedding = torch.randn(192, dtype=torch.float32) if embedding is None else _as_tensor(embedding, torch.float32).flatten()
            if self.augmentation_enabled:
                mel, token = self.augmentation(mel, token)
            cross = None
            if self.cross_sample_enabled and random.random() < self.cross_sample_prob:
                cross = self._get_random_prompt_mel(idx)
            text = None
            if sample.get('text_token', None) is not None:
                text = _as_tensor(sample['text_token'], torch.long).flatten()
            return {'speech_token': token, 'speech_feat': mel, 'embedding': embedding, 'cross_sample_mel': cross, 'text_token': text}
        except Exception as e:
            if idx < 3:
                print(f"Error loading sample {idx}: {e}  (keys: {list(sample.keys())})")
            return None


# ---------------------------------------------------------------------------------
# batch assembly (dataset.py:485-596)
# ---------------------------------------------------------------------------------
def _pad_stack(seqs: Sequence[torch.Tensor], value: float) -> torch.Tensor:
    n = max(s.shape[0] for s in seqs)
    out = seqs[0].new_full((len(seqs), n) + tuple(seqs[0].shape[1:]), value)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
    return out


def collate_fn(batch, max_feat_len_limit: Optional[int] = None):
    """Samples -> batch dict.  Utterances longer than `max_feat_len_limit` frames (default: config
    `max_feat_len`) are cut, and their speech / text tokens cut in the same proportion (floor)."""
    batch = [b for b in batch if b is not None]
    if not batch:
        return None
    if max_feat_len_limit is None:
        max_feat_len_limit = JOINT_TRAINING_CONFIG.get('max_feat_len', 150)
    for b in batch:
        n = b['speech_feat'].shape[0]
        if n > max_feat_len_limit:
            b['speech_feat'] = b['speech_feat'][:max_feat_len_limit]
            b['speech_token'] = b['speech_token'][:int(b['speech_token'].shape[0] * max_feat_len_limit / n)]
            if b.get('text_token') is not None:
                b['text_token'] = b['text_token'][:int(b['text_token'].shape[0] * max_feat_len_limit / n)]
    out = {
        'speech_token': _pad_stack([b['speech_token'] for b in batch], 0),
        'speech_token_len': torch.tensor([b['speech_token'].shape[0] for b in batch]),
        'speech_feat': _pad_stack([b['speech_feat'] for b in batch], MEL_PADDING_VALUE),
        'speech_feat_len': torch.tensor([b['speech_feat'].shape[0] for b in batch]),
        'embedding': torch.stack([b['embedding'] for b in batch]),
    }
    texts = [b.get('text_token', None) for b in batch]
    if all(t is not None for t in texts):                 # the LLM branch needs text for every utterance of the batch
        out['text_token'] = _pad_stack(texts, 0)
        out['text_token_len'] = torch.tensor([t.shape[0] for t in texts])
    cross = [b.get('cross_sample_mel', None) for b in batch]
    if any(c is not None for c in cross):
        lens = [c.shape[0] if c is not None else 0 for c in cross]
        filled = [c if c is not None else torch.full((0, N_MELS), MEL_PADDING_VALUE) for c in cross]
        out['cross_sample_mel'] = _pad_stack(filled, MEL_PADDING_VALUE)
        out['cross_sample_mel_len'] = torch.tensor(lens)
    return out


class ShardSampler(Sampler[int]):
    """Data-parallel partition: every rank shuffles the SAME index list (seed + epoch), keeps the indices
    rank, rank + world, ... and only as many as make whole global batches -- so each optimiser step sees
    `world * batch_size` distinct utterances and no rank runs a step the others do not."""

    def __init__(self, n: int, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0, shuffle: bool = True):
        self.n, self.batch_size, self.rank, self.world, self.seed, self.shuffle = n, batch_size, rank, world, seed, shuffle
        self.epoch = 0
        self.steps = n // (batch_size * world)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.steps * self.batch_size

    def __iter__(self) -> Iterator[int]:
        order = list(range(self.n))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        order = order[:self.steps * self.batch_size * self.world]
        return iter(order[self.rank::self.world])


def create_dataloader(data_dir: str, batch_size: int = 2, num_workers: int = 0, max_duration: float = 15.0,
                      rank: int = 0, world: int = 1, seed: int = 0) -> DataLoader:
    """Reference behaviour at world == 1 (shuffle, drop_last, pinned memory); rank-strided shards otherwise."""
    dataset = FlowFinetuneDataset(data_dir=data_dir, max_duration=max_duration)
    if world > 1:
        sampler = ShardSampler(len(dataset), batch_size, rank, world, seed)
        return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers, collate_fn=collate_fn,
                          drop_last=True, pin_memory=torch.cuda.is_available())
    return DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers, collate_fn=collate_fn,
                      drop_last=True, pin_memory=torch.cuda.is_available())
