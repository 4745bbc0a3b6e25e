"""Child process of tests/test_trainer_gpu.py::test_data_parallel_trainer_equals_global_batch: one data-parallel rank
(or the single-process global-batch run when WORLD_SIZE=1) of the REAL trainer stack -- JointLLMFlowModel + FlatAdamW +
Trainer.fit with gradient accumulation -- on its shard of a ragged global batch.  Writes history + final LoRA tensors."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402


def main():
    out_path, use_graph = sys.argv[1], int(sys.argv[2])
    from cosyvoice_lora_finetune_framework_amd import dp
    rank, local, world = dp.init_from_env()          # before any GPU call (CVFT_SINGLE_DEVICE=1: every rank on GPU 0)
    torch.cuda.set_device(local)
    from conftest import load_json
    from helpers import build_flow_product, build_llm_product
    from cosyvoice_lora_finetune_framework_amd.llm_flow_model import JointLLMFlowModel
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics
    from cosyvoice_lora_finetune_framework_amd.synthetic import cfm_draws, synth_batch
    from cosyvoice_lora_finetune_framework_amd.train_joint import JointLightningModule, Trainer
    meta = load_json("tiny_meta.json")
    big = os.environ.get("CVFT_DPTEST_BIG") == "1"       # the driver's N > 1 configuration in small: rank-16 adapters (slab
    if big:                                               # products + LoraGradSink), shards of 8 (two Flow chains per rank)
        for k in ("flow", "llm"):
            meta[k]["lora"]["r"], meta[k]["lora"]["alpha"] = 16, 32
    num = Numerics(dtype=torch.float32)
    jm = JointLLMFlowModel(build_llm_product(meta["llm"], "cuda", num), build_flow_product(meta["flow"], "cuda", num), 'joint',
                           llm_loss_weight=2.0, flow_loss_weight=1.0)
    module = JointLightningModule('joint', learning_rate=1e-3, min_lr=1e-5, warmup_steps=2, weight_decay=0.01, model=jm,
                                  numerics=num)
    # 4 global batches of 4 ragged utterances; every rank's shard (utterances rank::world) holds one utterance of the
    # global maximum lengths, because the reference's length regulator interpolates the PADDED batch (Lt_max -> T_max):
    # a shard padded to other maxima would not reproduce the global-batch run
    G = [([24, 24, 17, 20], [7, 7, 5, 6], [13, 13, 9, 11]), ([22, 22, 15, 19], [6, 6, 4, 5], [12, 12, 8, 10]),
         ([24, 24, 21, 13], [7, 7, 3, 6], [13, 13, 11, 7]), ([20, 20, 18, 11], [5, 5, 5, 2], [11, 11, 10, 6])]
    if big:          # 16 utterances per global batch: every pattern of G four times, the two maxima first (ranks 0 and 1 get one each)
        G = [(fl[:2] + (fl[2:] * 7), tl[:2] + (tl[2:] * 7), kl[:2] + (kl[2:] * 7)) for fl, tl, kl in G]
    batches, draws = [], []
    for i, (fl, tl, kl) in enumerate(G):
        full = synth_batch(fl, text_lens=tl, token_lens=kl, seed=500 + i, text_vocab=100, speech_vocab=50)
        d = cfm_draws(len(fl), max(fl), 900 + i)
        rows = list(range(rank, len(fl), world))
        batches.append({k: (v[rows] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == len(fl) else v) for k, v in full.items()})
        draws.append({k: v[rows] for k, v in d.items()})
    # every all-reduce of the run, with the thread that issued it: the test asserts that the loss-denominator exchange never
    # runs on the main thread (i.e. in front of a step's launches / replay), VERDICT round 3 item 8
    import threading
    import torch.distributed as dist
    calls = []
    if world > 1:
        orig = dist.all_reduce

        def recording_all_reduce(t, *a, **k):
            calls.append((threading.current_thread() is threading.main_thread(), t.numel(), t.device.type, k.get("group") is not None))
            return orig(t, *a, **k)
        dist.all_reduce = recording_all_reduce
    tr = Trainer(max_epochs=2, accumulate_grad_batches=2, gradient_clip_val=1.0, train_mode=False, log_every_n_steps=1,
                 save_checkpoints=False, use_graph=bool(use_graph), draws_fn=lambda ep, bi, b: draws[bi])
    tr.fit(module, batches)
    torch.save({"history": tr.history, "world": world, "rank": rank, "graph_stats": tr.graph_stats, "allreduce_calls": calls,
                "params": {k: v.detach().cpu() for k, v in jm.named_parameters() if v.requires_grad}}, out_path)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
