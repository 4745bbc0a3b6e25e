#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units), corrected as
MI355X_MICROARCH.md section HBM prescribes: on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 B => doubled;
WRITE_SIZE is exact for wide stores.   usage: pmc_traffic.py <fetch.csv> <write.csv> <kernel substring> <label> <out.json>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_lora_finetune_framework_amd.build_id import csrc_sha16


def avg(path, counter, sub):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and sub in r['Kernel_Name']]
    return sum(v) / len(v), len(v)


f, nf = avg(sys.argv[1], 'FETCH_SIZE', sys.argv[3])
w, nw = avg(sys.argv[2], 'WRITE_SIZE', sys.argv[3])
out = {"kernel": sys.argv[4], "csrc_sha16": csrc_sha16(), "mangled_contains": sys.argv[3], "launches_sampled": [nf, nw], "FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w,
       "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python bench.py --graph 0`; "
                 "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 B (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)"}
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(out)
