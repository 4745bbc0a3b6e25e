#!/usr/bin/env python3
"""Per-(kernel, grid) medians of rocprofv3 --pmc counters (counter_collection.csv), one column per counter."""
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0]
            if 'attn' not in name and len(sys.argv) > 0 and 'ALL' not in sys.argv:
                continue
            key = (name[:44], int(r['Grid_Size']) // 256 if 'Grid_Size' in r else 0)
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in agg.values() for c in v})
print(f"{'kernel':46s} {'blocks':>7s} " + " ".join(f"{n[-18:]:>18s}" for n in names))
for k, v in sorted(agg.items()):
    med = lambda x: sorted(x)[len(x) // 2] if x else float('nan')
    print(f"{k[0]:46s} {k[1]:7d} " + " ".join(f"{med(v.get(n, [])):18.0f}" for n in names))
