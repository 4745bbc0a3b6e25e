#!/usr/bin/env python3
"""Per-kernel breakdown of the LAST training step in a rocprofv3 --kernel-trace csv (steps are delimited by the
once-per-step lora_grad_reduce_kernel)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) for r in rows]
ev.sort()
idx = [i for i, e in enumerate(ev) if 'lora_grad_reduce' in e[2]]
step = ev[idx[-2] + 1:idx[-1] + 1]
print(f"step span {(step[-1][1] - step[0][0]) / 1e6:.2f} ms, {len(step)} kernels")
d = collections.defaultdict(list)
for s, e, n, g in step:
    key = n[:58]
    if 'gemm_glds' in n:
        key = 'glds ' + n[21:60]
    d[key].append((e - s) / 1e3)
print(f"sum of kernel durations {sum(sum(v) for v in d.values()) / 1e3:.2f} ms")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k:60s} n={len(v):4d} mean={sum(v) / len(v):7.1f} us  total={sum(v) / 1e3:6.2f} ms")
