#!/usr/bin/env python3
"""Per-kernel hardware counters of the training step: MFMA-busy fraction of the matrix-core kernels and HBM-side GB/s of
the memory-bound kernels, against the gfx950 peaks (BASELINE.json north_star: "rocprof HBM GB/s and MFMA-busy counters
reported against gfx950 peak").

  collect (on the GPU box; three rocprofv3 --pmc passes, the program directly after `--`, no other trace domain):
      python tools/counters.py collect gpurun_out/counters_r2
  summarise (anywhere):
      python tools/counters.py summarise gpurun_out/counters_r2 profiles/r2_counters.json

Definitions (MI355X_MICROARCH.md):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the
               counter sums the 8 XCDs), i.e. the share of the chip's matrix-pipe cycles the launch kept busy.
  hbm_gbps   = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B / duration: FETCH_SIZE tallies 128-B read requests at 64 B on
               gfx950 (doubled), WRITE_SIZE is exact for wide stores; Infinity-Cache hits are counted too, so this is
               memory-side traffic, an upper bound on DRAM traffic.  Peak 8 000 GB/s (spec), ~6 300 achievable.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CMD = ["python3", "bench.py", "--steps", "3", "--warmup", "2", "--via-trainer", "0", "--graph", "0", "--no-cpu-baseline",
       "--no-roofline"]
PASSES = {"mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"],
          "fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"]}
PEAK_HBM_GBPS, SIMDS = 8000.0, 1024


def short(name: str) -> str:
    n = name.split("(")[0]
    for a, b in (("void ", ""), ("_Z16gemm_glds_kernelILi", "gemm_glds<"), ("_Z16gemm_p256_kernelI", "gemm_p256<"), ("_Z11gemm_kernelIDF16b", "gemm<bf16,")):
        n = n.replace(a, b)
    return n[:72]


def collect(out: str):
    os.makedirs(out, exist_ok=True)
    for tag, ctrs in PASSES.items():
        d = os.path.join(out, tag)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "--", *CMD]
        print("+", " ".join(cmd), flush=True)
        with open(os.path.join(out, tag + ".log"), "w") as lf:
            subprocess.run(cmd, check=True, stdout=lf, stderr=subprocess.STDOUT)


def load(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
                rows[k]["_ns_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return rows


def summarise(out: str, dst: str):
    m, fe, wr = load(os.path.join(out, "mfma")), load(os.path.join(out, "fetch")), load(os.path.join(out, "write"))
    res = {}
    for k, c in m.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        if n == 0:
            continue
        cyc = sum(c["GRBM_GUI_ACTIVE"]) / 8.0
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        ns = sum(c["_ns_GRBM_GUI_ACTIVE"])
        e = {"launches_sampled": n, "avg_us": ns / n / 1e3, "total_ms_sampled": ns / 1e6,
             "mfma_busy": busy / (SIMDS * cyc) if cyc > 0 else 0.0, "clock_ghz": cyc / ns if ns > 0 else 0.0,
             "valu_per_mfma": (sum(c["SQ_INSTS_VALU"]) / sum(c["SQ_INSTS_MFMA"])) if sum(c["SQ_INSTS_MFMA"]) > 0 else None}
        f, w = fe.get(k), wr.get(k)
        if f and w and f["FETCH_SIZE"] and w["WRITE_SIZE"]:
            fb = sum(f["FETCH_SIZE"]) / len(f["FETCH_SIZE"]) * 1024.0 * 2.0
            wb = sum(w["WRITE_SIZE"]) / len(w["WRITE_SIZE"]) * 1024.0
            dur = (sum(f["_ns_FETCH_SIZE"]) / len(f["FETCH_SIZE"]) + sum(w["_ns_WRITE_SIZE"]) / len(w["WRITE_SIZE"])) / 2.0
            e.update(read_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_gbps=(fb + wb) / dur,
                     hbm_frac_of_8tbps=(fb + wb) / dur / PEAK_HBM_GBPS)
        res[k] = e
    top = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_ms_sampled"]))
    from cosyvoice_lora_finetune_framework_amd.build_id import csrc_sha16
    doc = {"command": " ".join(CMD), "csrc_sha16": csrc_sha16(), "method": __doc__.split("Definitions")[1].strip(), "kernels": top}
    json.dump(doc, open(dst, "w"), indent=1)
    print(f"{'kernel':74s} {'n':>5s} {'avg us':>8s} {'mfma_busy':>9s} {'VALU/MFMA':>9s} {'GB/s':>8s} {'of 8TB/s':>8s}")
    for k, e in list(top.items())[:40]:
        print(f"{k:74s} {e['launches_sampled']:5d} {e['avg_us']:8.1f} {e['mfma_busy']:9.3f} "
              f"{(e['valu_per_mfma'] or 0):9.1f} {e.get('hbm_gbps', 0):8.0f} {e.get('hbm_frac_of_8tbps', 0):8.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        summarise(sys.argv[2], sys.argv[3])
