#!/usr/bin/env python3
"""What runs between the join of one step's chains and the fork of the next (rocprofv3 --kernel-trace csv): the kernels from the last
block_* / attention launch of a step to the first GEMM-or-gather of the next, with start offsets and durations.
usage: step_tail.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows)
idx = [i for i, e in enumerate(ev) if 'adamw_flat' in e[2]]
i = idx[-2]
lo = max(0, i - 40)
t0 = ev[lo][0]
for s, e, n, q in ev[lo:i + 60]:
    print(f"+{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  q{q}  {n[:110]}")
