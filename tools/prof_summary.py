#!/usr/bin/env python3
"""Print a per-kernel table from a rocprofv3 --kernel-trace --stats run (kernel_stats.csv)."""
import csv, glob, sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"{f}: total {tot/1e6:.1f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:88]:88s} n/step={float(r['Calls'])/steps:7.0f} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
