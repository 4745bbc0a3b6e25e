#!/bin/bash
# Diagnostic libraries of the 256x256 GEMM (never the product build): libcvft_p256dbgN.so with -DP256_DBG=N (gemm_p256.hip:
# 1 no DMA in the k-loop, 2 fragment reads in the first k-tile only, 3 no MFMAs, 4 no barriers).  Results are garbage; run
# tools/bench_p256.py time under CVFT_LIB_PATH=<that library> beside the product build in ONE gpurun call.
set -e
cd "$(dirname "$0")/../cosyvoice_lora_finetune_framework_amd/csrc"
bash build.sh > /dev/null
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DP256_DBG=$n -c gemm_p256.hip -o build/gemm_p256_dbg$n.o
  objs=$(ls build/*.o | grep -v "gemm_p256\|_stamps")
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcvft_p256dbg$n.so $objs build/gemm_p256_dbg$n.o
  echo "built $(realpath ../libcvft_p256dbg$n.so)"
done
