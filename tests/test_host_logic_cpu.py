"""Host-side logic that needs no GPU: LR schedule, LM index maps / targets, LoRA injection + export
key contract, data-parallel reductions over gloo (world size 2)."""
import math
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import load_json


def test_lr_schedule_matches_reference_run():
    """train_joint.py:211-216 arithmetic vs the LambdaLR values recorded from the reference run."""
    from cosyvoice_lora_finetune_framework_amd.optim import lr_lambda
    log = load_json("train_tiny_log.json")
    hp = log["hp"]
    steps = [r for r in log["log"] if "lr" in r]
    for i, r in enumerate(steps):
        lr = hp["lr"] * lr_lambda(i, hp["warmup"], log["total_steps"], hp["min_lr"], hp["lr"])
        assert abs(lr - r["lr"]) <= 1e-12 + 1e-7 * r["lr"], (i, lr, r["lr"])
    assert lr_lambda(0, 50, 1000, 1e-6, 2e-4) == 0.0
    assert lr_lambda(10 ** 6, 50, 1000, 1e-6, 2e-4) >= 1e-6 / 2e-4


def test_lm_index_maps_match_oracle_layout():
    """llm.py:88-95 + llm_flow_model.py:129-139 as one host-built gather map."""
    from oracle import ref_math as R
    from cosyvoice_lora_finetune_framework_amd.llm_model import TransformerLM
    B, Lx, Lt = 3, 6, 9
    text_len, sp_len = torch.tensor([6, 2, 4]), torch.tensor([9, 5, 1])
    sp = torch.randint(0, 50, (B, Lt))
    idx, tgt, lens, L = TransformerLM.build_index_maps(text_len, sp_len, sp, B, Lx, Lt, eos=50)
    ref_t = R.build_lm_target(text_len, sp, sp_len, 50)
    assert L == ref_t.shape[1] and torch.equal(tgt.view(B, L).long(), ref_t)
    assert lens.tolist() == [3 + 6 + 9, 3 + 2 + 5, 3 + 4 + 1]
    # emulate the gather on the host and compare with the oracle's ragged concat
    d = 4
    special, spk, enc, semb = torch.randn(2, d), torch.randn(B, d), torch.randn(B * Lx, d), torch.randn(B * Lt, d)
    src = torch.cat([special, spk, enc, semb])
    got = torch.where(idx.view(-1, 1) >= 0, src[idx.clamp(min=0).long()], torch.tensor(-1.0)).view(B, L, d)
    for i in range(B):
        a, b = int(text_len[i]), int(sp_len[i])
        exp = torch.cat([special[0:1], spk[i:i + 1], enc[i * Lx:i * Lx + a], special[1:2], semb[i * Lt:i * Lt + b]])
        assert torch.equal(got[i, :3 + a + b], exp)
        assert bool((got[i, 3 + a + b:] == -1).all())


def test_lora_injection_and_export_contract(tiny_meta):
    """apply_lora_to_model stats / freezing (lora.py:134-227) and merged-key contract (lora.py:284-323) on CPU
    parameter containers (no compute)."""
    from cosyvoice_lora_finetune_framework_amd.flow_model import build_flow_model
    from cosyvoice_lora_finetune_framework_amd import lora
    fm = tiny_meta["flow"]
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in fm["build"].items()}
    m = build_flow_model(None, 'cpu', **kw)
    assert sorted(m.state_dict().keys()) == fm["base_keys"]
    st = lora.apply_lora_to_model(m, r=4, lora_alpha=8, lora_dropout=0.0, target_modules=fm["lora"]["targets"])
    assert st == fm["stats"]
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == {k: s for k, s in fm["spec"]}
    assert all(('lora_' in n) == p.requires_grad for n, p in m.named_parameters())
    wrapped = m.decoder.estimator.mid_blocks[0][1][0].attn1.to_q
    assert isinstance(wrapped, lora.LoRALinear) and wrapped.scaling == 2.0
    assert float(wrapped.lora_B.abs().sum()) > 0          # normal(0, .01), not zeros (lora.py:60-62)
    w0 = wrapped.original_layer.weight.clone()
    merged = lora.get_merged_state_dict(m)
    assert sorted(merged.keys()) == fm["base_keys"]
    delta = wrapped.lora_B @ wrapped.lora_A * 2.0
    assert torch.allclose(merged["decoder.estimator.mid_blocks.0.1.0.attn1.to_q.weight"], w0 + delta, atol=1e-6)
    sd = lora.get_lora_state_dict(m)
    assert len(sd) == 2 * st["replaced_layers"]


def test_lora_conv1d_wrapper_keys():
    from cosyvoice_lora_finetune_framework_amd import lora

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = nn.Conv1d(8, 12, 1)
            self.other = nn.Conv1d(8, 8, 3, padding=1)
    net = Net()
    st = lora.apply_lora_to_model(net, r=2, lora_alpha=4, target_modules=['to_q'])
    assert st["replaced_layers"] == 1 and isinstance(net.to_q, lora.LoRAConv1d)
    assert {"to_q.lora_A.weight", "to_q.lora_B.weight", "to_q.original_layer.weight"} <= set(net.state_dict())
    merged = lora.get_merged_state_dict(net)
    assert set(merged) == {"to_q.weight", "to_q.bias", "other.weight", "other.bias"}


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cosyvoice_lora_finetune_framework_amd import dp
    r, _, w = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    theta = torch.randn(5, requires_grad=True)          # "LoRA parameters", identical on all ranks
    # ragged global batch of 4 samples with different denominators; rank r owns samples 2r, 2r+1
    xs = torch.arange(20, dtype=torch.float32).view(4, 5) / 10.0
    dens = torch.tensor([3.0, 7.0, 2.0, 5.0])
    mine = list(dp.shard_indices(4, rank, world))
    num_local = sum(((xs[i] * theta).sum() ** 2) * dens[i] for i in mine)        # sum over "frames"
    den_local = sum(dens[i] for i in mine)
    local_mean = num_local / den_local
    wts = dp.loss_weights({"flow": float(den_local)}, torch.device("cpu"))
    (local_mean * wts["flow"]).backward()
    flat_g = theta.grad.clone()
    scale = dp.allreduce_flat_grads(flat_g)
    g_dp = flat_g * scale
    th2 = theta.detach().clone().requires_grad_(True)
    glob = sum(((xs[i] * th2).sum() ** 2) * dens[i] for i in range(4)) / dens.sum()
    glob.backward()
    m = dp.reduce_metrics(torch.tensor([float(num_local), float(den_local)]))
    ok = torch.allclose(g_dp, th2.grad, rtol=1e-5, atol=1e-6) and abs(float(m[0] / m[1]) - float(glob)) < 1e-5
    # dp.DenomExchange (what Trainer.fit uses): the same weights, exchanged by a second thread over a group of its own WHILE the
    # main thread runs collectives on the default group (the flat-gradient all-reduce of the step in flight); a rank without a
    # batch contributes zeros and still takes part
    import threading
    ex = dp.DenomExchange(["llm", "flow"])
    den = lambda i: {"flow": 10.0 * (rank + 1) + i, "llm": 0.0 if (i == 2 and rank == 1) else 3.0 + rank}
    side = []
    th = threading.Thread(target=lambda: side.extend(ex.weights(den(i)) for i in range(5)))
    th.start()
    for i in range(5):
        t = torch.ones(4)
        dist.all_reduce(t)
        ok = ok and float(t[0]) == world
    th.join()
    for i in range(5):
        lw = dp.loss_weights(den(i), torch.device("cpu"))
        want = torch.stack([torch.as_tensor(lw[k], dtype=torch.float32) for k in ex.keys])
        ok = ok and torch.allclose(side[i], want, rtol=1e-6, atol=0)
    ok = ok and ex.calls == 5 and ex.keys == ["flow", "llm"]
    q.put((rank, bool(ok)))
    dp.barrier()
    dist.destroy_process_group()


def test_data_parallel_equals_global_batch_gloo_world2():
    """DP with ragged shards: weighted local means + all-reduce(sum)/world == gradient of the global-batch
    mean (the reference's single-process loss), and the flat-bucket all-reduce plumbing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_mask_helpers_match_reference_fixtures():
    """utils.py helpers vs the reference's outputs (ops.npz)."""
    from conftest import load_npz
    from cosyvoice_lora_finetune_framework_amd import utils as U
    g = load_npz("ops.npz")
    assert torch.equal(U.make_pad_mask(torch.tensor([5, 3, 2])), g["pad_mask_5_3_2"])
    assert torch.equal(U.subsequent_chunk_mask(6, 1), g["chunk_mask_6_1"])
    assert torch.equal(U.mask_to_bias(torch.tensor([[True, False, True]]), torch.float32), g["mask_bias"])
    m = ~U.make_pad_mask(torch.tensor([4, 2]), 4).unsqueeze(1)
    cm = U.add_optional_chunk_mask(torch.zeros(2, 4, 8), m, False, False, 0, 1, -1)
    assert cm.shape == (2, 4, 4) and bool(cm[0].equal(torch.tril(torch.ones(4, 4, dtype=torch.bool))))
    assert cm[1, 3].tolist() == [True, True, False, False]
    assert U.pad_list([torch.ones(2), torch.ones(3)], -1).tolist() == [[1, 1, -1], [1, 1, 1]]


def test_callbacks_follow_reference_rules():
    """LossThresholdCallback: LLM checked before Flow, first hit returns (train_joint.py:80-102); EarlyStopping:
    min_delta 1e-3, patience."""
    from cosyvoice_lora_finetune_framework_amd.train_joint import EarlyStopping, LossThresholdCallback

    class T:
        callback_metrics = {}
        should_stop = False
    t = T()
    cb = LossThresholdCallback(llm_loss_threshold=1.5, flow_loss_threshold=0.3)
    t.callback_metrics = {"llm_loss_epoch": 1.6, "flow_loss_epoch": 0.31}
    cb.on_train_epoch_end(t)
    assert not t.should_stop
    t.callback_metrics = {"llm_loss_epoch": 1.6, "flow_loss_epoch": 0.29}
    cb.on_train_epoch_end(t)
    assert t.should_stop
    t2 = T()
    es = EarlyStopping(patience=2)
    for v in (1.0, 0.9, 0.8995, 0.8999):
        t2.callback_metrics = {"train_loss_epoch": v}
        es.on_train_epoch_end(t2)
    assert t2.should_stop


def test_data_path_matches_reference_dataset():
    """parquet shard -> samples -> batch dict.  The fixture holds what the reference's dataset.py
    produced from tests/golden/data_shard under random.seed(7) / torch.manual_seed(7) (tools/make_golden.py gen_data):
    data.list resolution of a foreign absolute path, flattened-mel decoding, augmentation draw order, cross-sample
    prompts, proportional truncation, padding values, text_token only when every utterance has one."""
    import os
    import random
    from conftest import GOLD, load_npz
    from cosyvoice_lora_finetune_framework_amd import dataset as D
    g = load_npz("data_path.npz")
    d = os.path.join(GOLD, "data_shard")
    D.ANTI_LEAKAGE_CONFIG = {'cross_sample_enabled': True, 'cross_sample_prob': 0.5}
    ds = D.FlowFinetuneDataset(d, augmentation=True, verbose=False)
    ds.cross_sample_enabled, ds.cross_sample_prob = True, 0.5
    assert len(ds) == 6
    random.seed(7)
    torch.manual_seed(7)
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        for k, v in it.items():
            key = f"item{i}/{k}"
            if v is None:
                assert key not in g, key
            else:
                assert key in g and v.shape == g[key].shape, key
                assert torch.equal(v, g[key]) if v.dtype == torch.long else torch.allclose(v, g[key], atol=1e-6), key
    for name, idx, lim in (("b0", [0, 1, 2], 50), ("b1", [3, 4], 50), ("b2", [0, 5], 1000)):
        out = D.collate_fn([{k: (v.clone() if torch.is_tensor(v) else v) for k, v in items[i].items()} for i in idx], lim)
        ref = {k.split("/", 1)[1]: v for k, v in g.items() if k.startswith(name + "/")}
        assert set(out) == set(ref), (name, set(out) ^ set(ref))
        for k in ref:
            assert out[k].shape == ref[k].shape and out[k].dtype == ref[k].dtype, (name, k)
            assert torch.allclose(out[k].double(), ref[k].double(), atol=1e-6), (name, k)
    assert "text_token" in {k.split("/")[1] for k in g if k.startswith("b0/")} and "b1/text_token" not in g
    ds2 = D.FlowFinetuneDataset(d, augmentation=False, verbose=False)
    ds2.cross_sample_enabled = False
    it = ds2[2]
    for k, v in it.items():
        if v is not None:
            assert torch.allclose(v.double(), g[f"plain2/{k}"].double()), k


def test_shard_sampler_partitions_whole_global_batches():
    from cosyvoice_lora_finetune_framework_amd.dataset import ShardSampler
    parts = [list(ShardSampler(23, batch_size=2, rank=r, world=4, seed=5)) for r in range(4)]
    assert all(len(p) == 4 for p in parts)                       # 23 // (2*4) = 2 steps x 2 per rank
    flat = [i for p in parts for i in p]
    assert len(set(flat)) == len(flat) == 16
    s = ShardSampler(23, 2, 0, 4, seed=5)
    s.set_epoch(1)
    assert list(s) != parts[0]


def test_merge_joint_weights_from_trainer_checkpoint(tiny_meta, tmp_path):
    """merge_joint_weights.py:65-273: a trainer checkpoint (`model.llm.` / `model.flow.` prefixes, one foreign key, one
    shape-mismatched key) -> per-branch files in the ORIGINAL key names that load strictly into the un-wrapped models and
    hold W + (alpha/r) B A; the flow half equals the reference's own merged export (tests/golden/flow_tiny_merged.npz)."""
    import time
    from conftest import load_npz
    from helpers import build_flow_product, build_llm_product
    from cosyvoice_lora_finetune_framework_amd import merge_joint_weights as MJ
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    num = Numerics(dtype=torch.float32)
    trained = JointLLMFlowModel(build_llm_product(tiny_meta["llm"], "cpu", num), build_flow_product(tiny_meta["flow"], "cpu", num))
    sd = {f"model.{k}": v.clone() for k, v in trained.state_dict().items()}
    sd["model.flow.not_a_key"] = torch.zeros(3)
    sd["model.llm.llm_decoder.bias"] = torch.zeros(7)                           # wrong shape: skipped, base value kept
    ck = tmp_path / "joint_joint_epoch=1.ckpt"
    torch.save({"state_dict": sd, "epoch": 1}, ck)
    time.sleep(0.01)
    torch.save({"state_dict": {}}, tmp_path / "joint_flow_only_epoch=9.ckpt")
    assert MJ.find_latest_joint_checkpoint(str(tmp_path)) == str(ck)                  # prefers joint_joint over newer files
    assert MJ.find_latest_joint_checkpoint(str(tmp_path), "flow_only").endswith("joint_flow_only_epoch=9.ckpt")
    assert MJ.find_latest_joint_checkpoint(str(tmp_path), "llm_only") is None
    # a fresh model with different adapter values: everything must come from the checkpoint
    fresh = JointLLMFlowModel(build_llm_product(tiny_meta["llm"], "cpu", num), build_flow_product(tiny_meta["flow"], "cpu", num, seed=99))
    lo, fo = str(tmp_path / "llm.pt"), str(tmp_path / "flow.pt")
    MJ.merge_both_from_checkpoint(str(ck), lo, fo, model=fresh)
    flow_m, llm_m = torch.load(fo), torch.load(lo)
    assert not any("lora_" in k or "original_layer" in k for k in list(flow_m) + list(llm_m))
    gm = load_npz("flow_tiny_merged.npz")
    for k, v in gm.items():                                                          # the reference's merged tensors
        assert torch.allclose(flow_m[k], v, rtol=1e-6, atol=1e-7), k
    build_flow_product(tiny_meta["flow"], "cpu", num, lora=False).load_state_dict(flow_m, strict=True)
    build_llm_product(tiny_meta["llm"], "cpu", num, lora=False).load_state_dict(llm_m, strict=True)
    L = tiny_meta["llm"]["lora"]
    name = "llm.encoders.0.self_attn.linear_q"
    want = sd[f"model.llm.{name}.original_layer.weight"] + (L["alpha"] / L["r"]) * sd[f"model.llm.{name}.lora_B"] @ sd[f"model.llm.{name}.lora_A"]
    assert torch.allclose(llm_m[f"{name}.weight"], want, rtol=1e-6, atol=1e-7)
    assert not torch.equal(llm_m["llm_decoder.bias"], torch.zeros_like(llm_m["llm_decoder.bias"]))


def test_slab_plans_are_legal_for_the_rank_kernels():
    """LoraGradSink.plan / plan_deferred (host side of the deterministic LoRA-gradient slabs): rows per block a multiple of 32
    (the matrix-core slab kernel's k-step), slabs cover every row, and the block target is met whenever the shape allows."""
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    for M in (1, 31, 64, 4000, 5328, 8000, 9968, 100000):
        for Cn in (64, 256, 768, 1024, 4096):
            rpb, ns = HF.LoraGradSink.plan(M, Cn)
            assert rpb % 32 == 0 and ns * rpb >= M > (ns - 1) * rpb
            if rpb > 64:      # slabs of k * 128 rows run four waves per 64-column stripe (the stacked form of the slab kernel)
                assert rpb % 128 == 0 and -(-Cn // 64) * HF.SINK_STACK * ns >= HF.SINK_PLAN_BLOCKS
        rpb, ns = HF.LoraGradSink.plan_deferred(M)
        assert rpb % 32 == 0 and ns * rpb >= M > (ns - 1) * rpb


def test_loss_recombination_forms_agree():
    """The launch-saving recombination of the branch losses (llm_flow_model._scaled / _total, train_joint._weighted_total:
    unit factors skipped, no `0 + x`, one dot for the weighted sum) against the plain op-per-factor form: same value, same
    gradients, for one and two loss terms and with gradient accumulation."""
    from cosyvoice_lora_finetune_framework_amd import llm_flow_model as J
    from cosyvoice_lora_finetune_framework_amd import train_joint as TJ
    default = J.LOSS_FUSE
    try:
        for keys, accum in ((("llm", "flow"), 1), (("llm", "flow"), 4), (("flow",), 1), (("llm",), 2)):
            out = {}
            for fuse in (True, False):
                J.LOSS_FUSE = fuse
                leaves = {k: torch.tensor(1.5 + i, requires_grad=True) for i, k in enumerate(keys)}
                parts = {k: J._scaled(J._total([J._scaled(v * 0.25, 1.0), J._scaled(v * 0.75, torch.tensor(1.0))]), 2.0 if k == "llm" else 1)
                         for k, v in leaves.items()}
                w = torch.tensor([0.5, 1.25])
                total = TJ._weighted_total({f"{k}_loss": v for k, v in parts.items()}, keys, w, accum)
                total.backward()
                out[fuse] = (total.detach(), [leaves[k].grad.clone() for k in keys])
            assert torch.allclose(out[True][0], out[False][0], rtol=1e-6)
            for a, b in zip(out[True][1], out[False][1]):
                assert torch.allclose(a, b, rtol=1e-6)
    finally:
        J.LOSS_FUSE = default


def test_shard_sampler_reshuffles_per_epoch_and_ranks_stay_disjoint():
    """Trainer.fit calls sampler.set_epoch(epoch); every epoch is a new common shuffle, the ranks' shards are
    disjoint and together cover whole global batches only."""
    from cosyvoice_lora_finetune_framework_amd.dataset import ShardSampler
    n, bs, world = 37, 2, 3
    per_epoch = []
    for epoch in range(2):
        shards = []
        for rank in range(world):
            s = ShardSampler(n, bs, rank, world, seed=5)
            s.set_epoch(epoch)
            shards.append(list(s))
        flat = [i for sh in shards for i in sh]
        assert len(set(flat)) == len(flat) == (n // (bs * world)) * bs * world
        assert all(len(sh) == len(shards[0]) for sh in shards)
        per_epoch.append(shards)
    assert per_epoch[0] != per_epoch[1]


def test_trainer_denominators_have_a_fixed_key_set():
    """the all-reduced denominator vector has the same length on every rank whatever the local batch holds
    (None batch, or no text tokens): a missing term contributes 0."""
    import torch
    from cosyvoice_lora_finetune_framework_amd.train_joint import _batch_denoms
    full = {"speech_feat_len": torch.tensor([10, 7]), "speech_token_len": torch.tensor([5, 4]), "text_token": torch.zeros(2, 3)}
    no_text = {k: v for k, v in full.items() if k != "text_token"}
    for mode, keys in (("joint", {"flow", "llm"}), ("flow_only", {"flow"}), ("llm_only", {"llm"})):
        for b in (full, no_text, None):
            assert set(_batch_denoms(b, mode)) == keys
    assert _batch_denoms(full, "joint") == {"flow": 17 * 80.0, "llm": 11.0}
    assert _batch_denoms(None, "joint") == {"flow": 0.0, "llm": 0.0} and _batch_denoms(no_text, "joint")["llm"] == 0.0


def test_trainer_fits_batches_to_captured_layouts():
    """Trainer._fit_layout (host side of the captured-step path): a batch joins the layout of an already captured step when
    that covers it within SHAPE_SLACK -- cheapest covering layout, tensors padded, exact maxima in `_true_dims` -- else
    keeps its own dims; once the graph budget is used up any covering layout within 2x is taken; and the LM index maps
    honour the joined step's LM length."""
    from cosyvoice_lora_finetune_framework_amd.llm_model import TransformerLM
    from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import LM_BUCKET, TEXT_BUCKET, Trainer
    batch = synth_batch([100, 80], text_lens=[9, 7], token_lens=[58, 40], seed=3, text_vocab=100, speech_vocab=50)
    T, Lt = batch['speech_feat'].shape[1], batch['speech_token'].shape[1]
    Lx = -(-batch['text_token'].shape[1] // TEXT_BUCKET) * TEXT_BUCKET
    L = -(-(9 + 58 + 3) // LM_BUCKET) * LM_BUCKET
    tr = Trainer(use_graph=False, max_graphs=4)
    fitted, dims = tr._fit_layout(batch)
    assert dims == (T, Lt, Lx, L, 2) and fitted['speech_feat'].shape[1] == T and fitted['text_token'].shape[1] == Lx
    assert fitted['_true_dims'].tolist() == [Lt, T, 50, 25, 13]
    tr._layouts += [(T + 30, Lt + 20, Lx, L + 16, 2),          # covers, but T is 30 % over: outside the slack
                    (T + 10, Lt + 6, Lx + 16, L + 16, 2),      # covers within the slack
                    (T + 4, Lt + 2, Lx, L, 2),                 # covers, cheapest
                    (T + 2, Lt - 1, Lx, L, 2),                 # does not cover Lt
                    (T + 1, Lt, Lx, L, 3)]                     # another batch size
    fitted, dims = tr._fit_layout(batch)
    assert dims == (T + 4, Lt + 2, Lx, L, 2)
    assert tuple(fitted['speech_feat'].shape) == (2, T + 4, 80) and fitted['speech_token'].shape[1] == Lt + 2
    assert torch.equal(fitted['speech_feat'][:, :T], batch['speech_feat']) and float(fitted['speech_feat'][:, T:].abs().max()) == 0.0
    assert fitted['_true_dims'].tolist() == [Lt, T, 50, 25, 13]
    tr._layouts = [(T + 30, Lt + 20, Lx, L + 16, 2)]
    assert tr._fit_layout(batch)[1] == (T, Lt, Lx, L, 2)       # only an out-of-slack layout: own dims, a new capture
    tr._graphs = {i: None for i in range(4)}                   # budget used up: the covering layout is taken anyway
    assert tr._fit_layout(batch)[1] == (T + 30, Lt + 20, Lx, L + 16, 2)
    # LM index maps of a batch that joined a step with a longer LM sequence
    idx, tgt, lm_len, L2 = TransformerLM.build_index_maps(batch['text_token_len'], batch['speech_token_len'], batch['speech_token'],
                                                          2, Lx, Lt, 50, LM_BUCKET, L + 16)
    assert L2 == L + 16 and idx.numel() == 2 * L2 and lm_len.tolist() == [70, 50] and int((idx.reshape(2, L2)[:, 70:] >= 0).sum()) == 0


def test_host_budget_splits_the_cores_between_local_ranks():
    """dp.host_budget: each of N local ranks gets cores // N of the process's cores, disjoint from the other
    ranks', and torch's intra-op pool is sized below the share (replay + prefetch threads keep a core each)."""
    import os
    import torch
    from cosyvoice_lora_finetune_framework_amd import dp
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity")
    before = os.sched_getaffinity(0)
    threads = torch.get_num_threads()
    try:
        n = len(before)
        shares = []
        for r in range(2):
            os.sched_setaffinity(0, before)
            hb = dp.host_budget(r, 2, pin=(n >= 4))
            shares.append(set(hb["cores"]))
            assert len(hb["cores"]) == max(1, n // 2) and 1 <= hb["torch_threads"] <= max(1, n // 2)
            if hb["pinned"]:
                assert os.sched_getaffinity(0) == set(hb["cores"])
        if n >= 2:
            assert not (shares[0] & shares[1])
        one = dp.host_budget(0, 1)
        assert not one["pinned"] or os.environ.get("CVFT_PIN_THREADS") == "1"
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)


def test_dropout_masks_of_two_sites_are_uncorrelated():
    """the counter-based masks (csrc/common.h cvft_keep4, replicated in tests/helpers.py) of two call sites, and of
    two steps, are independent draws -- not one table read at XOR-relabelled positions: the rate is p, the drop counts of aligned
    1024-element blocks at two sites / two seeds are uncorrelated, and no site's block counts equal another's sorted (a
    relabelling by XOR permutes aligned blocks among themselves: equal multisets of block counts)."""
    import numpy as np
    from helpers import drop_thr_host, keep_fields_host
    p, ng = 0.1, 1 << 18                                        # 2^20 elements
    thr = drop_thr_host(p)
    drops = {}
    for seed, site in ((1234, 7), (1234, 8), (1235, 7), (1234, 7 + 64)):
        f = keep_fields_host(seed, site, ng)
        d = (f < thr)
        assert abs(d.mean() - p) < 4 * np.sqrt(p * (1 - p) / d.size)            # rate
        assert abs(d[:, 0].mean() - d[:, 3].mean()) < 6 * np.sqrt(2 * p * (1 - p) / ng)      # fields of one draw alike
        drops[(seed, site)] = d.reshape(-1, 1024).sum(1).astype(np.float64)    # per aligned block of 1024 elements
    keys = list(drops)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            a, b = drops[keys[i]], drops[keys[j]]
            c = np.corrcoef(a, b)[0, 1]
            assert abs(c) < 5 / np.sqrt(a.size), (keys[i], keys[j], c)          # ~N(0, 1/sqrt(1024 blocks))
            assert not np.array_equal(np.sort(a), np.sort(b)), (keys[i], keys[j])
    # within one site: neighbouring elements independent (pair drop rate p^2)
    d = (keep_fields_host(1234, 7, ng) < thr).reshape(-1)
    both = (d[:-1] & d[1:]).mean()
    assert abs(both - p * p) < 5 * np.sqrt(p * p * (1 - p * p) / d.size)
