"""Joint LLM + Flow LoRA training loop -- the reference's train_joint.py semantics without
PyTorch-Lightning: same class names (``LossThresholdCallback``, ``JointLightningModule``), same
hyper-parameters and CLI flags (train_joint.py:58-242), same optimiser / scheduler / accumulation
/ clipping arithmetic (train_joint.py:198-226, 349-360), checkpoints in the Lightning key layout
(``state_dict`` with ``model.llm.`` / ``model.flow.`` prefixes, merge_joint_weights.py:95-104).

MI355X-first differences: one process per GPU (torch.distributed over RCCL when WORLD_SIZE > 1,
see dp.py), flat fp32 LoRA parameter/gradient buffers with a fused clip+AdamW kernel, bf16 MFMA
compute instead of fp16 autocast + GradScaler, and metrics kept on the device between log points
(no per-step host sync)."""
from __future__ import annotations

import argparse
import json
import math
import os
import time
from typing import Dict, Iterable, List, Optional

import torch

from . import dp
from .config import DATA_DIR, JOINT_TRAINING_CONFIG, MI355X_CONFIG, OUTPUT_DIR, PRETRAINED_MODEL_DIR, TRAIN_CONFIG
from .hipops.functional import LoraGradSink
from .modules import Numerics
from .optim import FlatAdamW, lr_lambda


class LossThresholdCallback:
    """train_joint.py:58-102: stop when an epoch-mean loss reaches its threshold; LLM is checked
    before Flow, and the first hit returns."""

    def __init__(self, llm_loss_threshold: Optional[float] = 2.0, flow_loss_threshold: Optional[float] = 0.3,
                 train_loss_threshold: Optional[float] = None, check_on_epoch_end: bool = True):
        self.llm_loss_threshold, self.flow_loss_threshold = llm_loss_threshold, flow_loss_threshold
        self.train_loss_threshold, self.check_on_epoch_end = train_loss_threshold, check_on_epoch_end

    def on_train_epoch_end(self, trainer, pl_module=None):
        if not self.check_on_epoch_end:
            return
        m = trainer.callback_metrics
        for key, thr, tag in (("llm_loss_epoch", self.llm_loss_threshold, "LLM"),
                              ("flow_loss_epoch", self.flow_loss_threshold, "Flow"),
                              ("train_loss_epoch", self.train_loss_threshold, "Total")):
            v = m.get(key)
            if v is not None and thr is not None and v <= thr:
                print(f"\n[{tag}] loss ({v:.4f}) reached threshold ({thr}); stopping")
                trainer.should_stop = True
                return


class EarlyStopping:
    """Lightning EarlyStopping(monitor='train_loss_epoch', min_delta=1e-3, patience=10, mode='min')
    as configured at train_joint.py:324-331."""

    def __init__(self, monitor: str = "train_loss_epoch", min_delta: float = 1e-3, patience: int = 10):
        self.monitor, self.min_delta, self.patience = monitor, min_delta, patience
        self.best, self.wait = math.inf, 0

    def on_train_epoch_end(self, trainer, pl_module=None):
        v = trainer.callback_metrics.get(self.monitor)
        if v is None:
            return
        if v < self.best - self.min_delta:
            self.best, self.wait = v, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                print(f"[EarlyStopping] {self.monitor} did not improve for {self.patience} epochs; stopping")
                trainer.should_stop = True


class JointLightningModule:
    """train_joint.py:105-226 (hyper-parameters, lazy model build, training_step, optimiser config)."""

    def __init__(self, training_mode: str = 'joint', learning_rate: float = 5e-5, min_lr: float = 1e-6,
                 warmup_steps: int = 200, weight_decay: float = 0.01, model=None, numerics: Optional[Numerics] = None,
                 pretrained_path: str = PRETRAINED_MODEL_DIR):
        self.training_mode, self.learning_rate, self.min_lr = training_mode, learning_rate, min_lr
        self.warmup_steps, self.weight_decay = warmup_steps, weight_decay
        self.model, self.numerics, self.pretrained_path = model, numerics or Numerics(), pretrained_path
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def setup(self, stage: Optional[str] = None):
        if self.model is None:
            from .llm_flow_model import build_joint_model
            self.model = build_joint_model(self.pretrained_path, device='cpu', training_mode=self.training_mode,
                                           llm_lora_config=JOINT_TRAINING_CONFIG.get('llm_lora'),
                                           flow_lora_config=JOINT_TRAINING_CONFIG.get('flow_lora'), numerics=self.numerics)
        self.model.to(self.device)
        self.model.set_numerics(self.numerics)

    def forward(self, batch, draws=None):
        return self.model(batch, self.device, draws)

    __call__ = forward

    def training_step(self, batch, batch_idx: int, draws=None):
        return self.forward(batch, draws)

    def configure_optimizers(self, max_grad_norm: float = 1.0) -> FlatAdamW:
        params = [p for p in self.model.parameters() if p.requires_grad]
        print(f"\ntrainable parameters: {sum(p.numel() for p in params):,}")
        return FlatAdamW(params, lr=self.learning_rate, weight_decay=self.weight_decay, betas=(0.9, 0.999),
                         max_grad_norm=max_grad_norm)

    def lr_at(self, step: int, total_steps: int) -> float:
        return self.learning_rate * lr_lambda(step, self.warmup_steps, total_steps, self.min_lr, self.learning_rate)


def _batch_denoms(batch) -> Dict[str, float]:
    d = {"flow": float(batch['speech_feat_len'].sum()) * 80.0}
    if 'speech_token_len' in batch:
        d["llm"] = float(batch['speech_token_len'].sum() + batch['speech_token_len'].numel())   # tokens + EOS
    return d


class Trainer:
    """The slice of pl.Trainer that train_joint.py:349-368 uses: max_epochs, accumulate_grad_batches,
    gradient_clip_val, callbacks, checkpoints (save_last + best), resume, step-level LR schedule."""

    def __init__(self, max_epochs: int = 100, accumulate_grad_batches: int = 1, gradient_clip_val: float = 1.0,
                 callbacks: Optional[list] = None, default_root_dir: str = OUTPUT_DIR, log_every_n_steps: int = 10,
                 draws_fn=None, save_checkpoints: bool = True, train_mode: bool = True):
        self.max_epochs, self.accum, self.clip = max_epochs, max(1, accumulate_grad_batches), gradient_clip_val
        self.train_mode = train_mode       # pl.Trainer.fit puts the module tree in .train() (dropouts active); False keeps the caller's mode
        self.callbacks = callbacks or []
        self.root, self.log_every, self.draws_fn, self.save_ckpt = default_root_dir, log_every_n_steps, draws_fn, save_checkpoints
        self.callback_metrics: Dict[str, float] = {}
        self.should_stop = False
        self.global_step = 0
        self.current_epoch = 0
        self.history: List[dict] = []
        self.rank, _, self.world = (0, 0, 1) if not torch.distributed.is_initialized() else \
            (torch.distributed.get_rank(), 0, torch.distributed.get_world_size())

    # -- checkpoint (Lightning key layout) ------------------------------------------------
    def _ckpt(self, module, opt):
        sd = {f"model.{k}": v.detach().cpu() for k, v in module.model.state_dict().items()}
        return {"state_dict": sd, "optimizer": opt.state_dict(), "epoch": self.current_epoch,
                "global_step": self.global_step, "hyper_parameters": dict(training_mode=module.training_mode,
                learning_rate=module.learning_rate, min_lr=module.min_lr, warmup_steps=module.warmup_steps,
                weight_decay=module.weight_decay)}

    def save_checkpoint(self, module, opt, name: str):
        if self.rank != 0 or not self.save_ckpt:
            return
        os.makedirs(self.root, exist_ok=True)
        torch.save(self._ckpt(module, opt), os.path.join(self.root, name))

    def load_checkpoint(self, module, opt, path: str):
        ck = torch.load(path, map_location="cpu")
        own = module.model.state_dict()
        for k, v in ck["state_dict"].items():
            kk = k[len("model."):] if k.startswith("model.") else k
            if kk in own:
                own[kk].copy_(v)
        opt.load_state_dict(ck["optimizer"])
        self.current_epoch, self.global_step = ck.get("epoch", 0), ck.get("global_step", 0)

    # -- fit ------------------------------------------------------------------------------
    def fit(self, module: JointLightningModule, dataloader, ckpt_path: Optional[str] = None):
        module.setup()
        if self.train_mode:
            module.model.train()
        dev = module.device
        nb = len(dataloader)
        total_steps = self.max_epochs * math.ceil(nb / self.accum)          # trainer.estimated_stepping_batches
        opt = module.configure_optimizers(self.clip)
        self.optimizer = opt
        if ckpt_path:
            self.load_checkpoint(module, opt, ckpt_path)
        logf = None
        if self.rank == 0 and self.save_ckpt:
            os.makedirs(self.root, exist_ok=True)
            logf = open(os.path.join(self.root, f"joint_{module.training_mode}_log.jsonl"), "a")
        best = math.inf
        keys = ("loss", "llm_loss", "flow_loss", "llm_acc")
        for epoch in range(self.current_epoch, self.max_epochs):
            self.current_epoch = epoch
            ep_sum = torch.zeros(len(keys), device=dev)
            ep_cnt = 0
            t0 = time.time()
            for bi, batch in enumerate(dataloader):
                if batch is None:                      # collate_fn: every sample of this batch failed to decode
                    continue
                draws = self.draws_fn(epoch, bi, batch) if self.draws_fn else None
                losses = module.training_step(batch, bi, draws)
                w = dp.loss_weights(_batch_denoms(batch), dev) if self.world > 1 else None
                if w is None:
                    total = losses['loss']
                else:
                    total = sum(losses[f"{k}_loss"] * w[k] for k in ("llm", "flow") if f"{k}_loss" in losses)
                with LoraGradSink():
                    (total / self.accum).backward()
                ep_sum += torch.stack([losses[k].detach().float() if k in losses else ep_sum.new_zeros(()) for k in keys])
                ep_cnt += 1
                rec = None
                if (bi + 1) % self.accum == 0 or bi + 1 == nb:
                    lr = module.lr_at(self.global_step, total_steps)
                    opt.set_lr(lr)
                    gscale = dp.allreduce_flat_grads(opt.flat_g)
                    opt.step(gscale)
                    if self.log_every and self.global_step % self.log_every == 0:
                        rec = dict(epoch=epoch, step=self.global_step, lr=lr, grad_norm=float(opt.grad_norm(gscale)),
                                   **{k: float(losses[k]) for k in keys if k in losses})
                    opt.zero_grad()
                    self.global_step += 1
                if rec is not None:
                    self.history.append(rec)
                    if logf:
                        logf.write(json.dumps(rec) + "\n")
                        logf.flush()
                if self.should_stop:
                    break
            means = dp.reduce_metrics(torch.cat([ep_sum, ep_sum.new_tensor([float(ep_cnt)])]))
            means = (means[:-1] / means[-1]).tolist()
            self.callback_metrics = {"train_loss_epoch": means[0], "train_loss": means[0]}
            if module.training_mode in ("joint", "llm_only"):
                self.callback_metrics.update(llm_loss_epoch=means[1], llm_acc_epoch=means[3])
            if module.training_mode in ("joint", "flow_only"):
                self.callback_metrics.update(flow_loss_epoch=means[2])
            if self.rank == 0:
                print(f"epoch {epoch}: " + "  ".join(f"{k}={v:.4f}" for k, v in self.callback_metrics.items()) +
                      f"  ({time.time() - t0:.1f}s)")
            self.save_checkpoint(module, opt, f"joint_{module.training_mode}_last.ckpt")
            if means[0] < best:
                best = means[0]
                self.save_checkpoint(module, opt, f"joint_{module.training_mode}_best.ckpt")
            for cb_ in self.callbacks:
                cb_.on_train_epoch_end(self, module)
            if self.should_stop:
                break
        if logf:
            logf.close()
        return self


class SyntheticLoader:
    """Deterministic synthetic batches in the reference batch format (synthetic.py); per-rank shard."""

    def __init__(self, n_batches: int, batch_size: int, T: int, seed: int = 1234, ragged: bool = False, rank: int = 0):
        self.n, self.bs, self.T, self.seed, self.ragged, self.rank = n_batches, batch_size, T, seed, ragged, rank

    def __len__(self):
        return self.n

    def __iter__(self):
        from .synthetic import synth_batch
        g = torch.Generator().manual_seed(self.seed + 7919 * self.rank)
        for i in range(self.n):
            lens = [self.T] * self.bs
            if self.ragged:
                lens = [int(self.T * (0.6 + 0.4 * float(torch.rand(1, generator=g)))) for _ in range(self.bs)]
                lens[0] = self.T
            yield synth_batch(lens, seed=self.seed + 1000 * self.rank + i)


def main():
    ap = argparse.ArgumentParser(description='LLM + Flow joint LoRA training (MI355X)')
    ap.add_argument('--mode', type=str, default='joint', choices=['joint', 'llm_only', 'flow_only'])
    ap.add_argument('--resume', type=str, default=None)
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--batch-size', type=int, default=None)
    ap.add_argument('--lr', type=float, default=None)
    ap.add_argument('--synthetic', type=int, default=0, help='train on N synthetic batches per epoch (no dataset needed)')
    ap.add_argument('--data-dir', type=str, default=None, help='directory with data.list / *.parquet (default: config.DATA_DIR)')
    ap.add_argument('--frames', type=int, default=500)
    ap.add_argument('--dtype', type=str, default=MI355X_CONFIG['compute_dtype'], choices=['bf16', 'fp32'])
    a = ap.parse_args()
    rank, local, world = dp.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    jc = JOINT_TRAINING_CONFIG
    epochs = a.epochs or jc['max_epochs']
    bs = a.batch_size or jc['batch_size']
    lr = a.lr or jc['learning_rate']
    num = Numerics(dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32)
    module = JointLightningModule(a.mode, learning_rate=lr, min_lr=TRAIN_CONFIG['min_learning_rate'],
                                  warmup_steps=TRAIN_CONFIG['warmup_steps'], weight_decay=TRAIN_CONFIG['weight_decay'],
                                  numerics=num)
    if a.synthetic > 0:
        loader = SyntheticLoader(a.synthetic, bs, a.frames, rank=rank, ragged=True)
    else:
        # parquet shards written by prepare_joint_data.py (train_joint.py:283-298); rank-strided shards under DP
        from .dataset import create_dataloader
        data_dir = a.data_dir or DATA_DIR
        if not os.path.isdir(data_dir):
            raise SystemExit(f"no dataset at {data_dir} (prepare_joint_data.py output); pass --data-dir or --synthetic N")
        loader = create_dataloader(data_dir, batch_size=bs, num_workers=0, rank=rank, world=world)
        if len(loader) == 0:
            raise SystemExit(f"{data_dir} holds fewer than {bs * world} usable utterances")
    trainer = Trainer(max_epochs=epochs, accumulate_grad_batches=jc['accumulate_grad_batches'],
                      gradient_clip_val=TRAIN_CONFIG['gradient_clip_val'],
                      callbacks=[EarlyStopping(), LossThresholdCallback(llm_loss_threshold=1.5, flow_loss_threshold=0.3)])
    trainer.fit(module, loader, ckpt_path=a.resume)
    if rank == 0:
        from .llm_flow_model import get_joint_merged_state_dict
        merged = get_joint_merged_state_dict(module.model)
        os.makedirs(OUTPUT_DIR, exist_ok=True)
        for k, sd in merged.items():
            torch.save(sd, os.path.join(OUTPUT_DIR, f"{k}_merged_{a.mode}.pt"))


if __name__ == "__main__":
    main()
