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
from . import llm_flow_model as _J
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


def _weighted_total(losses, keys, w, accum: int):
    """sum_k w[k] * losses[k + '_loss'] / accum with as few launches as it takes: these scalar ops sit between the last forward
    chain and the first backward launch with the whole chip waiting (each is ~5 us, and costs as much again in backward)."""
    from .llm_flow_model import LOSS_FUSE
    if not LOSS_FUSE:
        return sum(losses[f"{k}_loss"] * w[i] for i, k in enumerate(keys)) / accum
    if len(keys) == 1:
        t = losses[f"{keys[0]}_loss"] * w[0]
    else:
        t = torch.dot(torch.stack([losses[f"{k}_loss"].float() for k in keys]), w[:len(keys)].float())
    return t if accum == 1 else t / accum


def _fwd_bwd(model, batch, dev, draws, keys, w, accum: int):
    """One micro-step's forward + backward inside a LoraGradSink; the dict of detached loss scalars.  Chain by chain when the model
    offers it (JointLLMFlowModel.forward_backward, CVFT_CHAIN_BWD: no join of all chains between the two directions), else the
    reference's order: forward, weighted total, one backward."""
    from . import llm_flow_model as J
    if hasattr(model, "forward_backward") and J.chain_bwd_on(model):
        with LoraGradSink():
            losses = model.forward_backward(batch, dev, draws, w, accum)
        return {k: v.detach() for k, v in losses.items()}
    losses = model(batch, dev, draws)
    total = _weighted_total(losses, keys, w, accum)
    with LoraGradSink():
        total.backward()
    return {k: v.detach() for k, v in losses.items()}


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


def _batch_denoms(batch, training_mode: str = 'joint') -> Dict[str, float]:
    """Loss denominators of the local batch; the key set depends on the training mode only (never on the batch), so that
    every rank all-reduces a vector of the same length.  A missing batch / loss term contributes 0."""
    d: Dict[str, float] = {}
    if training_mode in ('joint', 'flow_only'):
        d["flow"] = float(batch['speech_feat_len'].sum()) * 80.0 if batch is not None else 0.0
    if training_mode in ('joint', 'llm_only'):
        ok = batch is not None and 'text_token' in batch
        d["llm"] = float(batch['speech_token_len'].sum() + batch['speech_token_len'].numel()) if ok else 0.0   # tokens + EOS
    return d


TEXT_BUCKET = 16      # graph path: text tokens padded to a multiple of this (masked by text_token_len)
LM_BUCKET = 16        # graph path: LM sequence length L rounded up to a multiple of this (masked by lm_len).  The padding rows are real
                      # work for every LM kernel: at 32, L = 333 became 352 (+5.7 % on the LLM branch, ~1 ms/step vs the pre-staged step)
# Mel frames T and speech tokens Lt: the reference's results depend on the padded batch dims themselves (the length regulator
# interpolates Lt_max -> T_max, length_regulator.py:44-50; the GroupNorms normalise over all T_max frames of the padded
# batch, modules.py:60-73), so they are never rounded up blindly.  A batch MAY be padded up to the layout of a step that is
# already captured when that costs at most SHAPE_SLACK: the exact maxima travel with the batch as device scalars
# (`_true_dims`) and the interpolation / GroupNorm kernels work from those (cvft.h `eff` / `t_eff`) -- same results, and a
# corpus whose batches rarely repeat a shape still runs on captured steps.
SHAPE_SLACK = float(os.environ.get("CVFT_SHAPE_SLACK", "0.125"))
T_ALIGN, LT_ALIGN = 4, 2
TRUE_DIM_LEVELS = 4   # `_true_dims` = [Lt_max, T_max, ceil(T_max / 2), ceil(T_max / 4), ceil(T_max / 8)]


class _Leaf:
    """placeholder of tensor number `i` in a packed batch tree"""
    __slots__ = ("i",)

    def __init__(self, i: int):
        self.i = i


def _tree_map(obj, fn_tensor, fn_leaf=None, memo=None):
    """rebuild a dict / list / tuple tree with tensors (or _Leaf placeholders) replaced; a tensor reachable twice (an
    unsplit branch's `_parts` entry aliases the batch) is mapped once"""
    memo = {} if memo is None else memo
    if torch.is_tensor(obj):
        if id(obj) not in memo:
            memo[id(obj)] = fn_tensor(obj)
        return memo[id(obj)]
    if isinstance(obj, _Leaf):
        return fn_leaf(obj)
    if isinstance(obj, dict):
        return {k: _tree_map(v, fn_tensor, fn_leaf, memo) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return tuple(_tree_map(v, fn_tensor, fn_leaf, memo) for v in obj)
    if isinstance(obj, list):
        return [_tree_map(v, fn_tensor, fn_leaf, memo) for v in obj]
    return obj


class _PackedBatch:
    """A prepared batch as ONE device byte slab plus a tree of typed views into it."""
    __slots__ = ("tree", "slab", "spec", "meta", "key", "dims")

    def bind(self, slab: torch.Tensor):
        """the same tree over another slab of this layout (the captured step's static copy)"""
        views = [slab[o:o + n].view(dt).view(shape) for (o, n, dt, shape) in self.meta]
        return _tree_map(self.spec, None, lambda leaf: views[leaf.i])


class _BatchPacker:
    """Host side of the graph path's input staging.  A prepared batch is ~40 small tensors (tokens, lengths, mel frames,
    LM index maps, the sub-batch copies of a split branch); moved one by one from pageable memory that was ~40 blit
    launches on a compute queue per step plus as many device-to-device copies into the captured step's static buffers,
    and with the step's three chains already on three hardware queues the extra queue traffic cost ~1 ms / step.  Here the
    tensors are laid out in ONE pinned host slab (256-byte aligned), which goes to the device as a single DMA on the copy
    stream; the static buffers of a captured step are views into one static slab, refreshed by one device copy."""
    ALIGN = 256
    ROUND = 1 << 16

    def __init__(self):
        import threading
        self.pool = {}                    # rounded byte size -> [(pinned buffer, event of its last upload)]
        self.lock = threading.Lock()      # pack() runs on the prefetch thread and, for a late re-fit, on the main thread

    def _pinned(self, nbytes: int):
        size = -(-nbytes // self.ROUND) * self.ROUND
        bufs = self.pool.setdefault(size, [])
        for ent in bufs:
            if ent[1] is None or ent[1].query():
                return ent
        ent = [torch.empty(size, dtype=torch.uint8).pin_memory(), None]
        bufs.append(ent)
        return ent

    def pack(self, tree, dev, stream) -> _PackedBatch:
        leaves, scalars = [], []

        def note(t):
            leaves.append(t.detach().contiguous())
            return _Leaf(len(leaves) - 1)

        def walk_scalars(o):             # ints / slices the captured step bakes in (LM length, sub-batch row ranges)
            if isinstance(o, dict):
                for k in sorted(o, key=str):
                    walk_scalars(o[k])
            elif isinstance(o, (tuple, list)):
                for v in o:
                    walk_scalars(v)
            elif isinstance(o, (int, slice)) and not isinstance(o, bool):
                scalars.append(repr(o))
        pk = _PackedBatch()
        pk.spec = _tree_map(tree, note)
        walk_scalars(pk.spec)
        off, meta = 0, []
        for t in leaves:
            n = t.numel() * t.element_size()
            meta.append((off, n, t.dtype, tuple(t.shape)))
            off += -(-max(n, 1) // self.ALIGN) * self.ALIGN
        total = max(off, self.ALIGN)
        with self.lock:                   # a pinned buffer is taken until the event of its upload is on the stream
            ent = self._pinned(total)
            host = ent[0]
            for t, (o, n, dt, shape) in zip(leaves, meta):
                if n:
                    host[o:o + n].view(dt).view(shape).copy_(t)
            with torch.cuda.stream(stream):
                pk.slab = torch.empty(total, dtype=torch.uint8, device=dev)
                pk.slab.copy_(host[:total], non_blocking=True)
                ent[1] = torch.cuda.Event()
                ent[1].record()
        pk.meta = meta
        pk.key = (tuple((shape, str(dt)) for (_, _, dt, shape) in meta), tuple(scalars))
        pk.tree = pk.bind(pk.slab)
        return pk


def _record_stream(obj, stream) -> None:
    """tensors allocated on the copy stream are consumed on `stream`: tell the caching allocator"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            _record_stream(v, stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)


class _Prefetcher:
    """Background thread that pulls the next batches from the dataloader and runs their host-side preparation (index
    maps, padding, host -> device copies on the copy stream) while the main thread is inside the current step: replaying
    a captured step blocks the host for most of the step (the launch queue is much shorter than the ~2 700-node graph),
    so preparation done on the main thread would add to the step time instead of hiding under it."""

    def __init__(self, iterable, fn, depth: int = 2):
        import queue
        import threading
        self.q = queue.Queue(maxsize=depth)
        self._end = object()
        self._stop = threading.Event()
        self._queue_mod = queue

        dev_index = torch.cuda.current_device() if torch.cuda.is_available() else None

        def put(item) -> bool:
            """blocking put that gives up when the consumer has closed the prefetcher"""
            while not self._stop.is_set():
                try:
                    self.q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                if dev_index is not None:       # a new thread starts on device 0: bind it to this rank's GPU (one process per GPU)
                    torch.cuda.set_device(dev_index)
                for item in iterable:
                    if self._stop.is_set() or not put(fn(item)):
                        return
                put(self._end)
            except BaseException as e:          # forwarded to the consumer
                put(e)
        self.t = threading.Thread(target=work, daemon=True)
        self.t.start()

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is self._end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        """Stop the worker (early stop, an exception in the step, KeyboardInterrupt): it would otherwise stay blocked in
        q.put forever, holding the dataloader iterator, prepared device slabs and pinned buffers, and keep issuing
        copy-stream work during checkpoint save / teardown."""
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except self._queue_mod.Empty:
            pass
        self.t.join(timeout=30)


GRAPH_OPT = os.environ.get('CVFT_GRAPH_OPT', '1') != '0'      # the optimiser step inside the captured micro-step (_StepGraph)
_TIMING = [] if os.environ.get('CVFT_TRAINER_TIMING') else None      # diagnostic: per-replay (events, host stamps), read by bench.py


class _StepGraph:
    """One captured micro-step -- forward + backward into the flat LoRA-gradient buffer -- for one batch shape.  The
    batch lives in static device buffers that the trainer refreshes before every replay; the per-rank loss weights (DP)
    and the injected CFM draws are static inputs too."""

    def __init__(self, module, packed: _PackedBatch, draws, w: torch.Tensor, accum: int, flat_g: torch.Tensor, opt=None):
        """opt (a FlatAdamW, or None): the optimiser step -- clip, AdamW, bf16 shadows -- and the gradient reset are captured BEHIND
        the backward in the same graph (`self.steps_optimizer`).  Why: a captured step with parallel branches keeps the host inside
        hipGraphLaunch until the graph has all but finished (tools/dbg/replay_blocking.py: a three-branch graph's replay call
        returns after 5 ms of a 5.5 ms graph, a linear one's after 0.4 ms), so everything the host enqueues between two replays
        runs with the chip idle: the optimiser's six launches were 0.11 ms of every step.  Its learning rate and step count are
        device scalars (optim.FlatAdamW.lr_dev / step_dev), so the captured launches stay valid."""
        self.slab = packed.slab.clone()              # static inputs: one slab, the batch tree is views into it
        self.batch, self.draws, self.w = packed.bind(self.slab), draws, w
        model, dev = module.model, module.device
        keys = [k for k in ("llm", "flow") if (k == "llm" and module.training_mode in ('joint', 'llm_only')) or
                (k == "flow" and module.training_mode in ('joint', 'flow_only'))]
        self.steps_optimizer = opt is not None
        state = {"opt": False}

        def run():
            out = _fwd_bwd(model, self.batch, dev, self.draws, keys, self.w, accum)
            if state["opt"]:
                opt.step(1.0)
                opt.zero_grad()
            return out

        if self.steps_optimizer and opt.tiles is None:
            opt.refresh_shadows()                    # (builds the shadow tile table on the host: not inside a capture)
        saved = flat_g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            LoraGradSink.scattered = False
            run()                                   # allocator / pack warm-up off the capture
            if LoraGradSink.scattered:              # an adapter met more slab products than its buffer was sized for: the
                run()                               # next backward lays them out contiguously -- the layout capture will see
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the prefetch thread (allocations, host -> device copies on the copy stream) and, under DP, the
        # RCCL watchdog keep running during capture; in the default "global" mode any such call from another thread
        # invalidates the capture
        state["opt"] = self.steps_optimizer          # (the warm-up passes above ran without it)
        count = opt.step_count if self.steps_optimizer else 0
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = run()
        if self.steps_optimizer:
            opt.step_count = count                   # (step() counted the capture; replay() counts the steps that run)
        flat_g.copy_(saved)                          # the warm-up run accumulated once; capture itself executes nothing

    @staticmethod
    def _copy(dst, src, seen=None):
        """refresh static tensors one by one (the injected CFM draws; the batch itself is one slab copy)"""
        seen = set() if seen is None else seen
        if torch.is_tensor(dst):
            if id(dst) not in seen:
                seen.add(id(dst))
                dst.copy_(src, non_blocking=True)
        elif isinstance(dst, (tuple, list)):
            for a, b in zip(dst, src):
                _StepGraph._copy(a, b, seen)
        elif isinstance(dst, dict):
            for k in dst:
                _StepGraph._copy(dst[k], src[k], seen)

    def replay(self, packed: _PackedBatch, draws, w: torch.Tensor):
        if _TIMING is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            h0 = time.perf_counter()
        self.slab.copy_(packed.slab, non_blocking=True)
        if self.draws is not None:
            self._copy(self.draws, draws)
        self.w.copy_(w)
        if _TIMING is not None:
            ev[1].record()
            h1 = time.perf_counter()
        self.graph.replay()
        if _TIMING is not None:
            ev[2].record()
            ev.append((h0, h1, time.perf_counter()))
            _TIMING.append(ev)
        return self.out


class Trainer:
    """The slice of pl.Trainer that train_joint.py:349-368 uses: max_epochs, accumulate_grad_batches,
    gradient_clip_val, callbacks, checkpoints (save_last + best), resume, step-level LR schedule.

    On a GPU the micro-step (forward + backward) replays a captured hipGraph per batch layout (`use_graph`, default on;
    CVFT_TRAINER_GRAPH=0 or use_graph=False launches eagerly): a batch is padded up to the layout of an already captured
    step when one covers it within SHAPE_SLACK (exact maxima travel as device scalars: same results), else its own layout
    is captured (at most `max_graphs` at a time -- a captured step keeps its activations: ~7.6 GiB at B = 16, T ~ 500, so 16 of
    them are ~120 of the 288 GB; beyond that the least recently replayed step gives up its slot, `evict_after`, and only a
    corpus that cycles through more uncoverable layouts than slots runs some steps eagerly); the batch reaches the graph's static slab by one DMA + one device
    copy; all-reduce / clip / AdamW stay outside the graph."""

    def __init__(self, max_epochs: int = 100, accumulate_grad_batches: int = 1, gradient_clip_val: float = 1.0,
                 callbacks: Optional[list] = None, default_root_dir: str = OUTPUT_DIR, log_every_n_steps: int = 10,
                 draws_fn=None, save_checkpoints: bool = True, train_mode: bool = True, use_graph: Optional[bool] = None,
                 max_graphs: int = 16, on_step_end=None, evict_after: int = 32):
        self.max_epochs, self.accum, self.clip = max_epochs, max(1, accumulate_grad_batches), gradient_clip_val
        self.train_mode = train_mode       # pl.Trainer.fit puts the module tree in .train() (dropouts active); False keeps the caller's mode
        self.callbacks = callbacks or []
        self.root, self.log_every, self.draws_fn, self.save_ckpt = default_root_dir, log_every_n_steps, draws_fn, save_checkpoints
        self.callback_metrics: Dict[str, float] = {}
        self.should_stop = False
        self.global_step = 0
        self.current_epoch = 0
        self.history: List[dict] = []
        self.best = math.inf
        if use_graph is None:
            use_graph = os.environ.get("CVFT_TRAINER_GRAPH", "1") != "0"
        self.use_graph = bool(use_graph) and torch.cuda.is_available()
        self.max_graphs = max_graphs
        # When every capture slot is taken and NO captured step covers a batch, the least recently replayed step is retired and
        # the new layout captured in its place -- provided that step has not been replayed for `evict_after` micro-steps (a
        # capture costs ~4 eager steps: a corpus cycling through more uncoverable layouts than slots would otherwise pay one per
        # batch; with the age rule it pays at most one per `evict_after` steps and runs the rest eagerly, 4-5x a replay).
        self.evict_after = evict_after
        self._last_used: Dict[tuple, int] = {}
        self._micro_steps = 0
        self.on_step_end = on_step_end     # optional hook(trainer) after every optimiser step (bench.py --via-trainer)
        self.graph_stats = {"replays": 0, "eager": 0, "captures": 0}
        self._copy_stream = None
        self._packer = _BatchPacker()
        self._graphs: Dict[tuple, _StepGraph] = {}
        self._layouts: List[tuple] = []    # (T, Lt, text, LM length, B) of every captured step, read by the prefetch thread
        self._denoms = None                # dp.DenomExchange while fit() runs at world > 1
        self._term_keys: List[str] = []
        self.rank, _, self.world = (0, 0, 1) if not torch.distributed.is_initialized() else \
            (torch.distributed.get_rank(), 0, torch.distributed.get_world_size())

    # -- checkpoint (Lightning key layout) ------------------------------------------------
    def _ckpt(self, module, opt):
        from .hipops import functional as HF
        sd = {f"model.{k}": v.detach().cpu() for k, v in module.model.state_dict().items()}
        seed = HF.dropout_seed_state()          # rank-free: every rank re-applies its own offset on resume
        return {"state_dict": sd, "optimizer": opt.state_dict(), "epoch": self.current_epoch,
                "global_step": self.global_step, "best": self.best,
                "callbacks": [dict(type=type(c).__name__, **{k: v for k, v in vars(c).items() if isinstance(v, (int, float))})
                              for c in self.callbacks],
                "dropout_seed": seed,
                "hyper_parameters": dict(training_mode=module.training_mode,
                learning_rate=module.learning_rate, min_lr=module.min_lr, warmup_steps=module.warmup_steps,
                weight_decay=module.weight_decay)}

    def save_checkpoint(self, module, opt, name: str):
        if self.rank != 0 or not self.save_ckpt:
            return
        os.makedirs(self.root, exist_ok=True)
        torch.save(self._ckpt(module, opt), os.path.join(self.root, name))

    def load_checkpoint(self, module, opt, path: str):
        """Resume like Lightning: the stored epoch has finished, training continues with the next one; the best-loss
        mark, the callbacks' counters and the dropout seed continue too."""
        from .hipops import functional as HF
        ck = torch.load(path, map_location="cpu", weights_only=True)      # tensors + plain scalars / lists / dicts only
        own = module.model.state_dict()
        for k, v in ck["state_dict"].items():
            kk = k[len("model."):] if k.startswith("model.") else k
            if kk in own:
                own[kk].copy_(v)
        opt.load_state_dict(ck["optimizer"])
        self.current_epoch, self.global_step = ck.get("epoch", -1) + 1, ck.get("global_step", 0)
        self.best = ck.get("best", math.inf)
        for c, st in zip(self.callbacks, ck.get("callbacks", [])):
            if st.get("type") == type(c).__name__:
                for k, v in st.items():
                    if k != "type" and hasattr(c, k):
                        setattr(c, k, v)
        if ck.get("dropout_seed") is not None and torch.cuda.is_available():
            HF.set_dropout_seed_state(int(ck["dropout_seed"]))

    # -- one micro-step ---------------------------------------------------------------------
    def _prepare(self, module, batch):
        """Host side of a micro-step (runs on the prefetch thread): the data-parallel loss weights of this batch (world > 1),
        then the layout work of `_prepare_layout`.  Returns (batch, prepared, ready-event, weights)."""
        w = self._exchange_weights(module, batch)
        return self._prepare_layout(module, batch) + (w,)

    def _prepare_layout(self, module, batch):
        """bucket the text length, build the LM index maps, copy the batch to the device on the copy stream.  Returns
        (batch, prepared, ready-event).  No collective in here: `_micro_step` calls it again for a batch that was prepared
        before the step covering it was captured."""
        if batch is None or not (self.use_graph and hasattr(module.model, 'prepare_batch')):
            return batch, None, None
        dev = module.device
        fitted, dims = self._fit_layout(batch)
        # index maps / sub-batch split on the host, then ONE pinned slab -> ONE DMA on the copy stream (_BatchPacker): a
        # host -> device copy on the compute stream would wait for the whole previous step
        packed = self._packer.pack(module.model.prepare_batch(fitted, 'cpu', LM_BUCKET, dims[3]), dev, self._copy_stream)
        packed.dims = dims
        ev = torch.cuda.Event()
        with torch.cuda.stream(self._copy_stream):
            ev.record()
        return batch, packed, ev            # (draws_fn / loss denominators see the batch as the loader made it)

    def _exchange_weights(self, module, batch):
        """world > 1 (prefetch thread): this batch's loss weights (dp.DenomExchange -- a 2-float all-reduce over a gloo group
        of its own, one batch ahead of the step that uses them) as a device vector in `term_keys` order, copied on the copy
        stream from pinned memory; the step's stream waits for the batch's `ready` event or, on the eager path, for the event
        stored beside the vector.  None at world == 1."""
        if self._denoms is None:
            return None
        wh = self._denoms.weights(_batch_denoms(batch, module.training_mode))        # host floats, sorted-key order
        order = [self._denoms.keys.index(k) for k in self._term_keys]
        wh = wh[order].contiguous()
        dev = module.device
        if torch.device(dev).type != "cuda":
            return wh, None
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        with torch.cuda.stream(self._copy_stream):
            wd = wh.pin_memory().to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return wd, ev

    def _fit_layout(self, batch):
        """Shape side of the graph path: text length to its bucket, and (T, Lt, text, LM length) up to the layout of an
        already captured step when one covers the batch within SHAPE_SLACK.  Returns (padded batch with `_true_dims`,
        (T', Lt', Lx', L', B))."""
        feat, tok = batch['speech_feat'], batch['speech_token']
        B, T, Lt = feat.shape[0], feat.shape[1], tok.shape[1]
        has_text = 'text_token' in batch
        Lx = -(-batch['text_token'].shape[1] // TEXT_BUCKET) * TEXT_BUCKET if has_text else 0
        L = 0
        if has_text:
            L = int((batch['text_token_len'] + batch['speech_token_len']).max()) + 3          # llm_model.build_index_maps
            L = -(-L // LM_BUCKET) * LM_BUCKET
        best, cost = None, None
        full = len(self._graphs) >= self.max_graphs          # no capture left: any covering step beats the eager path (2.5x)
        for (T2, Lt2, Lx2, L2, B2) in list(self._layouts):
            if B2 != B or T2 < T or Lt2 < Lt or Lx2 < Lx or L2 < L:
                continue
            if not full and (T2 > T * (1 + SHAPE_SLACK) or Lt2 > Lt * (1 + SHAPE_SLACK) + 4 or L2 > L * (1 + SHAPE_SLACK) + LM_BUCKET):
                continue
            if full and T2 > 2 * T:
                continue
            c = (T2 / T) * (max(L2, 1) / max(L, 1))
            if cost is None or c < cost:
                best, cost = (T2, Lt2, Lx2, L2, B2), c
        # nothing covers it: its own layout (frames to a multiple of 4, tokens of 2: near-equal maxima share one capture)
        T2, Lt2, Lx2, L2, _ = best if best is not None else (-(-T // T_ALIGN) * T_ALIGN, -(-Lt // LT_ALIGN) * LT_ALIGN, Lx, L, B)
        out = dict(batch)
        pad = torch.nn.functional.pad
        if T2 > T:
            out['speech_feat'] = pad(feat, (0, 0, 0, T2 - T))
        if Lt2 > Lt:
            out['speech_token'] = pad(tok, (0, Lt2 - Lt))
        if has_text and Lx2 > batch['text_token'].shape[1]:
            out['text_token'] = pad(batch['text_token'], (0, Lx2 - batch['text_token'].shape[1]))
        out['_true_dims'] = torch.tensor([Lt] + [-(-T // (1 << l)) for l in range(TRUE_DIM_LEVELS)], dtype=torch.int32)
        return out, (T2, Lt2, Lx2, L2, B)

    def _micro_step(self, module, opt, batch, draws, w, prepared=None, ready=None):
        """forward + backward of one local batch; returns the dict of detached loss scalars.  `w` = per-term loss weights
        (device tensor, ones at world == 1)."""
        dev = module.device
        keys = [k for k in ("llm", "flow") if (k == "llm" and module.training_mode in ('joint', 'llm_only')) or
                (k == "flow" and module.training_mode in ('joint', 'flow_only'))]
        if prepared is not None:
            main = torch.cuda.current_stream()
            if self._layouts and prepared.dims not in self._layouts and self._fit_layout(batch)[1] != prepared.dims:
                # prepared (by the prefetch thread) before the step that covers it was captured: fit it again
                _, prepared, ready = self._prepare_layout(module, batch)
            if draws is not None:
                T2 = prepared.dims[0]
                with torch.cuda.stream(self._copy_stream):
                    draws = {k: v.to(dev) for k, v in draws.items()}
                    if 'z' in draws and draws['z'].shape[-1] < T2:       # injected noise follows the padded frame count
                        draws['z'] = torch.nn.functional.pad(draws['z'], (0, T2 - draws['z'].shape[-1]))
                    ready = torch.cuda.Event()
                    ready.record()
            main.wait_event(ready)
            prepared.slab.record_stream(main)         # allocated on the copy stream, consumed on `main`
            _record_stream(draws, main)
            # layout of the slab (every tensor's shape / dtype) + the Python scalars the step bakes in (LM length L,
            # sub-batch row ranges) + whether CFM draws are injected
            key = (prepared.key, draws is not None and tuple(sorted(draws)))
            self._micro_steps += 1
            g = self._graphs.get(key)
            if g is None:
                # capture at first sight: a layout only gets here when no captured step covers it (_fit_layout)
                if 0 < self.max_graphs <= len(self._graphs):
                    self._retire_oldest()
                if len(self._graphs) < self.max_graphs:
                    g = self._graphs[key] = _StepGraph(module, prepared, draws, w.clone(), self.accum, opt.flat_g,
                                                       opt if self._graph_steps_optimizer(module) else None)
                    g.dims = prepared.dims
                    self._layouts.append(prepared.dims)
                    if os.environ.get('CVFT_TRAINER_DEBUG'):
                        print(f"[trainer] captured layout {prepared.dims} (exact {tuple(batch['speech_feat'].shape)}, {tuple(batch['speech_token'].shape)})", flush=True)
                    self.graph_stats["captures"] += 1
            if g is not None:
                self._last_used[key] = self._micro_steps
                self.graph_stats["replays"] += 1
                self._optimizer_ran = g.steps_optimizer
                if g.steps_optimizer:
                    opt.step_count += 1
                return g.replay(prepared, draws, w)
            batch = prepared.tree
        self.graph_stats["eager"] += 1
        return _fwd_bwd(module.model, batch, dev, draws, keys, w, self.accum)

    def _graph_steps_optimizer(self, module) -> bool:
        """the captured step ends with the optimiser step when every micro-step is an optimiser step, no other rank's gradients are
        waited for and nobody asked to see the gradients first (CVFT_GRAPH_OPT=0: never)"""
        return (GRAPH_OPT and self.accum == 1 and self.world == 1 and getattr(module, "on_before_optimizer_step", None) is None)

    def _retire_oldest(self):
        """Free the capture slot of the least recently replayed step if it is old enough (see `evict_after`): its graph, its
        static batch slab and the activations its private pool holds go back to the allocator.  The LoRA slab workspaces and
        reduce tables it wrote to are never freed (other captured steps may share them)."""
        key = min(self._graphs, key=lambda k: self._last_used.get(k, 0))
        if self._micro_steps - self._last_used.get(key, 0) < self.evict_after:
            return
        torch.cuda.synchronize()                  # no replay of it may still be in flight
        g = self._graphs.pop(key)
        self._last_used.pop(key, None)
        if g.dims in self._layouts and not any(o.dims == g.dims for o in self._graphs.values()):
            self._layouts.remove(g.dims)
        del g
        self.graph_stats["retired"] = self.graph_stats.get("retired", 0) + 1

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
        keys = ("loss", "llm_loss", "flow_loss", "llm_acc")
        term_keys = [k for k in ("llm", "flow") if k in _batch_denoms(None, module.training_mode)]   # order of _micro_step
        ones = torch.ones(len(term_keys), device=dev)
        self._term_keys = term_keys
        # world > 1: the denominators of batch i + 1 are exchanged by the prefetch thread over a gloo group of its own while
        # batch i's step replays -- no collective in front of a replay (dp.DenomExchange; created here: new_group is collective)
        self._denoms = dp.DenomExchange(sorted(term_keys)) if self.world > 1 else None
        for epoch in range(self.current_epoch, self.max_epochs):
            self.current_epoch = epoch
            sampler = getattr(dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):           # DP shards: a new common shuffle every epoch (ShardSampler)
                sampler.set_epoch(epoch)
            ep_sum = torch.zeros(len(keys), device=dev)
            ep_cnt = 0
            t0 = time.time()
            losses = None
            if self.use_graph and self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream()
            prefetch = _Prefetcher(dataloader, lambda b: self._prepare(module, b))
            try:
                for bi, (batch, prepared, ready, wx) in enumerate(prefetch):
                    # A rank whose batch failed to decode (collate_fn -> None) still takes part in every collective of the
                    # step with a zero-weight contribution: no rank ever skips an all-reduce the others enter (the prefetch
                    # thread has already exchanged this batch's denominators: _exchange_weights).
                    self._optimizer_ran = False
                    w = ones
                    if wx is not None:
                        w, w_ready = wx
                        if w_ready is not None:
                            torch.cuda.current_stream().wait_event(w_ready)
                            w.record_stream(torch.cuda.current_stream())
                    if batch is not None:
                        draws = self.draws_fn(epoch, bi, batch) if self.draws_fn else None
                        _J._chain_event(('step begin',))            # (diagnostic stamps: CVFT_CHAIN_EVENTS, llm_flow_model.py)
                        self._optimizer_ran = False
                        if self.use_graph and self._graph_steps_optimizer(module):
                            opt.set_lr(module.lr_at(self.global_step, total_steps))      # (a captured step may end with the optimiser's)
                        losses = self._micro_step(module, opt, batch, draws, w, prepared, ready)
                        _J._chain_event(('micro-step done',))
                        ep_sum += torch.stack([losses[k].float() if k in losses else ep_sum.new_zeros(()) for k in keys])
                        ep_cnt += 1
                    rec = None
                    if (bi + 1) % self.accum == 0 or bi + 1 == nb:
                        lr = module.lr_at(self.global_step, total_steps)
                        gscale = 1.0
                        if not self._optimizer_ran:          # (else: the captured step just ran clip + AdamW + shadows + the reset)
                            opt.set_lr(lr)
                            gscale = dp.allreduce_flat_grads(opt.flat_g)
                            hook = getattr(module, "on_before_optimizer_step", None)      # (Lightning's module hook of that name)
                            if hook is not None:
                                hook(opt)
                            opt.step(gscale)
                        if self.log_every and self.global_step % self.log_every == 0 and losses is not None:
                            rec = dict(epoch=epoch, step=self.global_step, lr=lr, grad_norm=float(opt.grad_norm(gscale)),
                                       **{k: float(losses[k]) for k in keys if k in losses})
                        if not self._optimizer_ran:
                            opt.zero_grad()
                        _J._chain_event(('optimiser done',))
                        self.global_step += 1
                        if self.on_step_end is not None:
                            self.on_step_end(self)
                    if rec is not None:
                        self.history.append(rec)
                        if logf:
                            logf.write(json.dumps(rec) + "\n")
                            logf.flush()
                    if self.should_stop:
                        break
            finally:
                prefetch.close()
            means = dp.reduce_metrics(torch.cat([ep_sum, ep_sum.new_tensor([float(ep_cnt)])]))
            means = (means[:-1] / means[-1].clamp_min(1.0)).tolist()      # (an epoch with no decodable batch: zeros, not NaN)
            self.callback_metrics = {"train_loss_epoch": means[0], "train_loss": means[0]}
            if module.training_mode in ("joint", "llm_only"):
                self.callback_metrics.update(llm_loss_epoch=means[1], llm_acc_epoch=means[3])
            if module.training_mode in ("joint", "flow_only"):
                self.callback_metrics.update(flow_loss_epoch=means[2])
            if self.rank == 0:
                print(f"epoch {epoch}: " + "  ".join(f"{k}={v:.4f}" for k, v in self.callback_metrics.items()) +
                      f"  ({time.time() - t0:.1f}s)")
            self.save_checkpoint(module, opt, f"joint_{module.training_mode}_last.ckpt")
            if means[0] < self.best:
                self.best = means[0]
                self.save_checkpoint(module, opt, f"joint_{module.training_mode}_best.ckpt")
            for cb_ in self.callbacks:
                cb_.on_train_epoch_end(self, module)
            if self.should_stop:
                break
        if logf:
            logf.close()
        return self


class SyntheticLoader:
    """Deterministic synthetic batches in the reference batch format (synthetic.py); per-rank shard.  `cache=True` builds
    the epoch's batches once (a dataset already decoded in host memory, what DataLoader workers hand over): the CPU
    random-number generation of a 16 x 500 x 80 batch costs more than a whole GPU step."""

    def __init__(self, n_batches: int, batch_size: int, T: int, seed: int = 1234, ragged=False, rank: int = 0,
                 cache: bool = False):
        self.n, self.bs, self.T, self.seed, self.ragged, self.rank = n_batches, batch_size, T, seed, ragged, rank
        self._cache = list(self._gen()) if cache else None

    def __len__(self):
        return self.n

    def _gen(self):
        from .synthetic import synth_batch
        g = torch.Generator().manual_seed(self.seed + 7919 * self.rank)
        for i in range(self.n):
            lens = [self.T] * self.bs
            if self.ragged:
                lens = [int(self.T * (0.6 + 0.4 * float(torch.rand(1, generator=g)))) for _ in range(self.bs)]
                if self.ragged != 2:          # (2: no utterance pinned to T -- every batch has its own T_max / Lt_max)
                    lens[0] = self.T
            yield synth_batch(lens, seed=self.seed + 1000 * self.rank + i)

    def __iter__(self):
        return iter(self._cache) if self._cache is not None else self._gen()


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
