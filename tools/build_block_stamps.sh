#!/bin/bash
# Diagnostic library with in-kernel stamps in the fused block-tail kernels (never the product build): libcvft_bfstamps.so
set -e
cd "$(dirname "$0")/../cosyvoice_lora_finetune_framework_amd/csrc"
bash build.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DBF_STAMPS -c block_fused.hip -o build/block_fused_stamps.o
objs=$(ls build/*.o | grep -v block_fused | grep -v _stamps)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcvft_bfstamps.so $objs build/block_fused_stamps.o
echo "built $(realpath ../libcvft_bfstamps.so)"
