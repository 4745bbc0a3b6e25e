#!/usr/bin/env python3
"""Per-shape kernel durations of the attention kernels from a rocprofv3 --kernel-trace CSV (tools/bench_attn.py run)."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'attn' not in r['Kernel_Name']:
        continue
    key = (r['Kernel_Name'].split('(')[0][:44], int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], r['Grid_Size_Z'], r['VGPR_Count'],
           r['LDS_Block_Size'])
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print(f"{k[0]:46s} grid {k[1]:2d}x{k[2]:>2s}x{k[3]:>2s} vgpr {k[4]:>3s} lds {k[5]:>6s} n {len(v):4d} med {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
