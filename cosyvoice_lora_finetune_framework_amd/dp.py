"""Data-parallel replicas over RCCL/xGMI: one process per GPU, frozen weights replicated,
ONE all-reduce(sum) of the flat LoRA-gradient buffer per optimiser step (skipped on
accumulation micro-steps -- DDP ``no_sync`` semantics, vendored cosyvoice/utils/executor.py:64-65),
plus a 2-float exchange of loss denominators so that ragged batches reproduce the reference's
single-process *global-batch* means (flow_matching.py:192; label_smoothing_loss.py:91-96).

The reference itself trains with devices=1 (train_joint.py:352); nothing here is ported.
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests)."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world).  Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set by torch.distributed.run."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL (read when the HIP runtime initialises)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CVFT_SINGLE_DEVICE"):       # rehearsal: several ranks sharing GPU 0 (use with CVFT_DIST_BACKEND=gloo)
        local = 0
    if world > 1 and not dist.is_initialized():
        # this rank's share of the host first: the threads the process group and torch create inherit it
        # local world: LOCAL_WORLD_SIZE, else the node's GPU count capped by the world (device_count does not initialise
        # the GPU on this stack); the share is only taken when this rank's place on the node is known (LOCAL_RANK set):
        # a rank-only launcher would otherwise pin every rank to slice 0
        if "LOCAL_WORLD_SIZE" in os.environ:
            lw = int(os.environ["LOCAL_WORLD_SIZE"])
        else:
            ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            lw = min(world, ndev) if ndev > 0 else world
        if "LOCAL_RANK" in os.environ:
            host_budget(local % max(lw, 1), lw)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CVFT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            try:        # bind the communicator to this rank's GPU up front (no device guessing in barrier / first collective)
                dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
            except TypeError:
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def host_budget(local_rank: int, local_world: int, pin: bool = None) -> dict:
    """Per-rank share of the host for one-process-per-GPU runs on one node: the calling thread (and every thread it starts
    afterwards: torch's intra-op pool, the trainer's prefetch thread, the collective backend's workers) is restricted to cores
    [local_rank * per, (local_rank + 1) * per) of the affinity mask, per = cores // local_world, and torch's intra-op pool is
    sized to per - 2 (<= 16): replaying the ~2 000-node step graph occupies one host core for ~9 ms of every 23 ms step and the
    prefetch thread another; eight ranks left to the scheduler with 16 intra-op threads each (round 2) oversubscribe the host.
    Threads are NOT pinned one per core: a pinned main thread hands its one-core mask to every pool it creates later (measured on
    the two-rank rehearsal: the step went from 107 to 235 ms).  pin = None: only when local_world > 1 (CVFT_PIN_THREADS=0/1
    overrides).  Returns {"cores": [...], "torch_threads": n, "pinned": bool}."""
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    env = os.environ.get("CVFT_PIN_THREADS")
    if pin is None:
        pin = (local_world > 1) if env is None else env != "0"
    per = max(1, len(cores) // max(local_world, 1))
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    nthreads = max(1, min(per - 2, 16))
    if pin and hasattr(os, "sched_setaffinity") and len(mine) >= 2:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            pin = False
    torch.set_num_threads(nthreads)
    return {"cores": mine, "torch_threads": nthreads, "pinned": bool(pin)}


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_indices(n: int, rank: int, world: int, drop_last: bool = True) -> range:
    """Contiguous slice of a (same-seed-shuffled) index list for this rank."""
    per = n // world if drop_last else -(-n // world)
    return range(rank * per, min(n, (rank + 1) * per))


def loss_weights(local_denoms: Dict[str, float], device) -> Dict[str, float | torch.Tensor]:
    """For each loss term return world * den_local / sum_ranks(den): scaling the local *mean* loss by it
    before backward makes all-reduce(sum)/world of the gradients equal the gradient of the
    global-batch mean.  Uniform shards -> 1.0.  Device tensors, no host sync."""
    keys = sorted(local_denoms)
    if world_size() == 1:
        return {k: 1.0 for k in keys}
    t = torch.tensor([float(local_denoms[k]) for k in keys], dtype=torch.float32)
    if torch.device(device).type == "cuda":
        # pinned + non_blocking: a pageable host -> device copy synchronises the stream, i.e. the host would wait for the whole
        # previous step before it may start enqueueing the next one (the captured step takes the host ~10 ms to enqueue)
        t = t.pin_memory().to(device, non_blocking=True)
    else:
        t = t.to(device)
    tot = t.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    w = t * world_size() / tot.clamp_min(1e-12)
    return {k: w[i] for i, k in enumerate(keys)}


class DenomExchange:
    """The loss-denominator exchange of `loss_weights`, taken OFF the step's critical path: the trainer's prefetch thread
    calls `weights()` for batch i + 1 (host floats, a 2-float all-reduce over a gloo group of its own) while the main thread
    replays the captured step of batch i, so no collective -- i.e. no cross-rank rendez-vous -- sits on the compute stream in
    front of a replay.  Ranks must call `weights()` once per batch index, in order, batch present or not (a rank whose batch
    failed to decode contributes zeros), exactly like `loss_weights`.  Construct on the main thread (collective: new_group)."""

    def __init__(self, keys: Sequence[str]):
        self.keys = sorted(keys)
        self.world = world_size()
        self.group = dist.new_group(backend="gloo") if self.world > 1 else None      # CPU tensors whatever the main backend is
        self.calls = 0

    def weights(self, local_denoms: Dict[str, float]) -> torch.Tensor:
        """float32 HOST vector, one entry per key (sorted): world * den_local / sum_ranks(den); ones at world == 1."""
        assert sorted(local_denoms) == self.keys, (sorted(local_denoms), self.keys)
        t = torch.tensor([float(local_denoms[k]) for k in self.keys], dtype=torch.float32)
        if self.world == 1:
            return torch.ones_like(t)
        tot = t.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        return t * self.world / tot.clamp_min(1e-12)


def allreduce_flat_grads(flat_g: torch.Tensor) -> float:
    """Sum the flat LoRA gradient buffer over ranks (in place); returns the 1/world grad scale the
    optimiser applies inside its fused update kernel."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
    return 1.0 / w


def reduce_metrics(nums_dens: torch.Tensor) -> torch.Tensor:
    """all-reduce(sum) a small vector of [numerators..., denominators...] for logging global means
    (replaces Lightning's sync_dist=True scalar reduce, train_joint.py:152-159)."""
    if world_size() > 1:
        dist.all_reduce(nums_dens, op=dist.ReduceOp.SUM)
    return nums_dens


def barrier():
    if dist.is_initialized():
        dist.barrier()
