#!/usr/bin/env python3
"""Launches and kernel time per training step from a rocprofv3 --kernel-trace csv (steps are delimited by the once-per-step
lora_grad_reduce launch; the last N steps are averaged) -> profiles/*step_summary*.json, read by bench.py (roofline.step).
usage: step_summary.py <kernel_trace.csv> <out.json> <workload> <batch> <frames> [nsteps=5]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_lora_finetune_framework_amd.build_id import csrc_sha16

rows = list(csv.DictReader(open(sys.argv[1])))
out, workload, batch, frames = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
nsteps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# a step ends with the optimiser's adamw launch (one per step, after the reduce launches)
idx = [i for i, e in enumerate(ev) if 'adamw_flat' in e[2]]
assert len(idx) > nsteps + 1, "trace holds too few steps"
steps = [ev[idx[-k - 2] + 1: idx[-k - 1] + 1] for k in range(nsteps)]
launches = sum(len(s) for s in steps) / nsteps
kms = sum((e - s) for st in steps for s, e, _ in st) / nsteps / 1e6
span = sum((st[-1][1] - st[0][0]) for st in steps) / nsteps / 1e6
groups = collections.Counter()
gtime = collections.Counter()
for st in steps:
    for s, e, n in st:
        key = n.split('(')[0][:70]
        groups[key] += 1
        gtime[key] += (e - s)
top = [{"kernel": k, "launches_per_step": groups[k] / nsteps, "ms_per_step": gtime[k] / nsteps / 1e6}
       for k, _ in gtime.most_common(70)]
json.dump({"csrc_sha16": csrc_sha16(), "workload": workload, "batch": batch, "frames": frames, "steps_averaged": nsteps, "launches_per_step": launches,
           "kernel_ms_per_step": kms, "step_span_ms_under_profiler": span, "top_kernels": top,
           "note": "rocprofv3 --kernel-trace serialises nothing but adds per-dispatch overhead: the span is longer than the un-profiled step"},
          open(out, "w"), indent=1)
print(f"{workload} B={batch} T={frames}: {launches:.0f} launches/step, {kms:.2f} ms kernel time/step, span {span:.2f} ms")
