#!/usr/bin/env python3
"""Per-queue (= per chain) occupancy of the LAST training step in a rocprofv3 --kernel-trace csv: span, busy time, idle gaps,
and the kernels with the most time per queue.  Steps are delimited by the optimiser's adamw launch.
usage: chain_timeline.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows)
idx = [i for i, e in enumerate(ev) if 'adamw_flat' in e[2]]
step = ev[idx[-2] + 1: idx[-1] + 1]
t0, t1 = step[0][0], max(e[1] for e in step)
print(f"step span {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels")
by = collections.defaultdict(list)
for s, e, n, q, st in step:
    by[q].append((s, e, n))
for q, evs in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in evs)
    span = max(e for _, e, _ in evs) - min(s for s, _, _ in evs)
    gaps = [evs[i + 1][0] - evs[i][1] for i in range(len(evs) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"\nqueue {q}: {len(evs)} kernels, first +{(evs[0][0] - t0) / 1e6:.2f} ms, last end +{(max(e for _, e, _ in evs) - t0) / 1e6:.2f} ms, "
          f"busy {busy / 1e6:.2f} ms, span {span / 1e6:.2f} ms, gaps: {len(pos)} positive, total {sum(pos) / 1e6:.2f} ms, median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us")
    d = collections.Counter()
    c = collections.Counter()
    for s, e, n in evs:
        k = n.split('(')[0][:60]
        d[k] += e - s
        c[k] += 1
    for k, v in d.most_common(top):
        print(f"    {k:60s} n={c[k]:4d} total {v / 1e6:6.2f} ms  mean {v / c[k] / 1e3:6.1f} us")
