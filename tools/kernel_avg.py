#!/usr/bin/env python3
"""Calls and average duration of the kernels whose name contains any of the given substrings, from the kernel_stats.csv under a
rocprofv3 --kernel-trace --stats output directory.  usage: kernel_avg.py <dir> <substring> [...]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit(f"no *kernel_stats.csv under {sys.argv[1]} (rocprofv3 needs --output-format csv)")
for r in csv.DictReader(open(files[0])):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:7.1f} us')
