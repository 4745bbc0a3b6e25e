#!/usr/bin/env python3
"""Average duration per (kernel, grid) of the LAST third of a rocprofv3 kernel trace csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) * 2 // 3:]
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'][:60], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print(f"{k[0]:60s} grid {k[1]:6d} n {len(v):4d} mean {sum(v) / len(v):8.1f} us  min {min(v):8.1f}")
