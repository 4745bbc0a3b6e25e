#!/bin/bash
# Build libcvft.so for gfx950 in-tree (travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libcvft.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
mkdir -p build
pids=()
for f in core gemm gemm_glds gemm_p256 gemm_fp8 skinny lora_grad norm attention attn_mfma32 elementwise ce block_fused block_qkv block_lean block_wide block_wide8 block_qkv_wide; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_common.h -nt build/$f.o ] || [ attn_common.h -nt build/$f.o ] || [ block_common.h -nt build/$f.o ] || [ block_qkv_body.h -nt build/$f.o ] || [ ../../include/cvft.h -nt build/$f.o ]; then
    # attn_mfma32: MFMA results feed VALU softmax code, so keep them in VGPRs (no v_accvgpr copies)
    EXTRA=""; [ $f = attn_mfma32 ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
    hipcc $FLAGS $EXTRA -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT build/core.o build/gemm.o build/gemm_glds.o build/gemm_p256.o build/gemm_fp8.o build/skinny.o build/lora_grad.o build/norm.o build/attention.o build/attn_mfma32.o build/elementwise.o build/ce.o build/block_fused.o build/block_qkv.o build/block_lean.o build/block_wide.o build/block_wide8.o build/block_qkv_wide.o
echo "built $(realpath $OUT)"
