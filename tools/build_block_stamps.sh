#!/bin/bash
# Diagnostic library with in-kernel stamps in one family of chain kernels (never the product build): libcvft_bfstamps.so
# usage: build_block_stamps.sh [block_fused | block_lean]
set -e
which=${1:-block_fused}
cd "$(dirname "$0")/../cosyvoice_lora_finetune_framework_amd/csrc"
bash build.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DBF_STAMPS $EXTRA_DEFS -c $which.hip -o build/${which}_stamps.o
objs=""
for f in core gemm gemm_glds gemm_fp8 skinny lora_grad norm attention attn_mfma32 elementwise ce block_fused block_qkv block_lean block_wide block_wide8 block_qkv_wide; do
  [ $f = $which ] || objs="$objs build/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcvft_bfstamps.so $objs build/${which}_stamps.o
echo "built $(realpath ../libcvft_bfstamps.so)"
