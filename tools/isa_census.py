#!/usr/bin/env python3
"""Per-kernel instruction census of a csrc/*.hip file compiled for gfx950: 4-byte vs 8/16-byte global loads, sub-dword loads,
branches, scratch (spill) instructions, v_exp / v_rcp counts, VGPRs.  Short elementwise kernels whose launch time is their own
instruction stream show their problems here first (a run-time flag around a load keeps one scalar load and one branch per
element; an array indexed at run time goes to scratch).  CPU only (hipcc cross-compiles).
usage: isa_census.py <file.hip> [name substring ...]      e.g.  tools/isa_census.py cosyvoice_lora_finetune_framework_amd/csrc/norm.hip ln_bwd gn_fused"""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
pats = sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    stem = os.path.splitext(os.path.basename(src))[0]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.dirname(src),
           "-c", src, "-o", os.path.join(tmp, stem + ".o"), "-save-temps=obj"]
    if stem == "attn_mfma32":
        cmd += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    subprocess.run(cmd, check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
vg = {m.group(1): (m.group(2), m.group(3)) for m in
      re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", asm)}
parts = re.split(r"^(_Z\w+):[^\n]*\n", asm, flags=re.M)
print(f"{'kernel':64s} {'instr':>6s} {'ld4':>4s} {'ld8+':>4s} {'ld<4':>4s} {'br':>4s} {'scr':>4s} {'exp':>4s} {'rcp':>4s} {'vgpr':>5s} {'spill':>5s}")
for i in range(1, len(parts) - 1, 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if pats and not any(p in name for p in pats):
        continue
    n = lambda rx: len(re.findall(rx, body, flags=re.M))
    v, sp = vg.get(name, ("?", "?"))
    cols = [n(r"^\s+[a-z]"), n(r"global_load_dword\s"), n(r"global_load_dwordx[234]"), n(r"global_load_(?:u|s)(?:short|byte)|global_load_short"),
            n(r"s_cbranch"), n(r"scratch_"), n(r"v_exp_f32"), n(r"v_rcp_f32")]
    print(f"{name[:64]:64s} {cols[0]:6d} " + " ".join(f"{c:4d}" for c in cols[1:]) + f" {v:>5s} {sp:>5s}")
