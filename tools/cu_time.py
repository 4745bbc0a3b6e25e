#!/usr/bin/env python3
"""CU-time per kernel of the steady-state step from a rocprofv3 --kernel-trace csv: duration x the share of the chip's 256 CUs
a launch can hold at once (workgroups / workgroups-per-CU by its LDS, VGPR and thread footprint, capped at 256 CUs).  With three
chains sharing the chip the step time follows the SUM of CU-time: a latency-bound kernel on 63 CUs costs a
quarter of a chip-filling one of the same duration.   usage: cu_time.py <kernel_trace.csv> [nsteps=5] [top=30] [OUT.json workload batch frames]
(OUT.json: per-kernel kernel ms and chip-equivalent ms per step + the identity of the kernel sources; bench.py picks the kernel its
roofline record is about by chip ms)"""
import collections
import csv
import json
import os
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30


# workgroups one CU holds at once (csrc: dynamic LDS of the launch / __launch_bounds__ / registers)
RESIDENCY = [("gemm_p256", 1),                      # 160 KB ring
             ("block_tail", 1), ("block_qkv", 1), ("block_link", 1),   # 155 KB (32-row forms), 512 registers or >= 82 KB and 8 waves (64-row forms)
             ("gemm_glds_kernelILi128ELi128", 2),   # 64 KB, 8 waves
             ("gemm_glds_kernelILi96ELi256", 1),    # 135 KB
             ("gemm_glds_kernelILi64ELi64", 3),     # 48-64 KB ring, 4 waves
             ("attn32_bwd_dq_kernelILb1", 1),       # 123 KB (rel-pos windows)
             ("attn32_bwd_dq_rel2", 2),             # 80 896 bytes, 248 registers
             ("attn32_bwd_dkv_kernelILb1", 2),      # 71 KB, 336 registers -> one per SIMD and block
             ("attn32_bwd_fused", 2), ("attn32_fwd", 2),
             ("lora_rank_mfma", 4), ("gemm_kernelI", 4), ("skinny", 2)]


def gi(r, *names, default=0):
    for n in names:
        if n in r and r[n] not in ("", None):
            try:
                return int(float(r[n]))
            except ValueError:
                pass
    return default


ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = max(1, gi(r, 'Workgroup_Size_X', 'Workgroup_Size', default=256) * max(1, gi(r, 'Workgroup_Size_Y', default=1)) * max(1, gi(r, 'Workgroup_Size_Z', default=1)))
    grid = max(1, gi(r, 'Grid_Size_X', 'Grid_Size', default=wg) * max(1, gi(r, 'Grid_Size_Y', default=1)) * max(1, gi(r, 'Grid_Size_Z', default=1)))
    nwg = max(1, grid // wg)
    lds = gi(r, 'LDS_Block_Size', 'LDS_Block_Size_v', default=0)
    vg = gi(r, 'VGPR_Count', default=64) + gi(r, 'Accum_VGPR_Count', default=0)
    waves = -(-wg // 64)
    alloc = max(8, -(-vg // 8) * 8)
    w_simd = max(1, min(8, 512 // alloc))
    per_cu = max(1, min(160 * 1024 // lds if lds > 0 else 32, (w_simd * 4) // waves if waves <= w_simd * 4 else 1, 32 // waves if waves <= 32 else 1))
    # the trace reports STATIC LDS only (0 for every kernel here that sizes its LDS at launch) and the arch VGPRs without the
    # accumulators: residency of the kernels that matter comes from their sources (dynamic LDS bytes / registers), the rest by threads
    for sub, res in RESIDENCY:
        if sub in r['Kernel_Name']:
            per_cu = res
            break
    cus = min(256.0, nwg / per_cu) if nwg / per_cu >= 1 else max(1.0, float(nwg) / per_cu)
    ev.append((s, e, r['Kernel_Name'], cus, nwg, per_cu, (wg, grid, lds, vg)))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'adamw_flat' in e[2]]
assert len(idx) > nsteps + 1, "trace holds too few steps"
steps = [ev[idx[-k - 2] + 1: idx[-k - 1] + 1] for k in range(nsteps)]
kt, ct, n, cu_sum = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
first = {}
for st in steps:
    for s, e, name, cus, nwg, per_cu, raw in st:
        key = name.split('(')[0][:72]
        kt[key] += (e - s)
        ct[key] += (e - s) * cus / 256.0
        cu_sum[key] += cus
        n[key] += 1
        first.setdefault(key, (nwg, per_cu) + raw)
tot_k = sum(kt.values()) / nsteps / 1e6
tot_c = sum(ct.values()) / nsteps / 1e6
print(f"kernel time {tot_k:.2f} ms/step, chip-equivalent CU-time {tot_c:.2f} ms/step (= the step's floor if the CUs were never idle)")
print(f"{'kernel':72s} {'n/step':>7s} {'avg us':>8s} {'CUs':>6s} {'kern ms':>8s} {'chip ms':>8s} {'share':>6s}")
for key, v in ct.most_common(top):
    print(f"{key:72s} {n[key] / nsteps:7.1f} {kt[key] / n[key] / 1e3:8.1f} {cu_sum[key] / n[key]:6.0f} {kt[key] / nsteps / 1e6:8.3f} {v / nsteps / 1e6:8.3f} {100 * v / sum(ct.values()):5.1f}%  [first launch: wgs {first[key][0]}, per CU {first[key][1]}, threads {first[key][2]}, grid {first[key][3]}, lds {first[key][4]}, vgpr {first[key][5]}]")

if len(sys.argv) > 7:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cosyvoice_lora_finetune_framework_amd.build_id import csrc_sha16
    doc = {"csrc_sha16": csrc_sha16(), "workload": sys.argv[5], "batch": int(sys.argv[6]), "frames": int(sys.argv[7]), "steps_averaged": nsteps,
           "kernel_ms_per_step": tot_k, "chip_ms_per_step": tot_c,
           "kernels": [{"kernel": k, "launches_per_step": n[k] / nsteps, "ms_per_step": kt[k] / nsteps / 1e6, "chip_ms_per_step": v / nsteps / 1e6,
                        "cus": cu_sum[k] / n[k]} for k, v in ct.most_common()]}
    json.dump(doc, open(sys.argv[4], "w"), indent=1)
