#!/bin/bash
# Diagnostic library with in-kernel stamps in the attention forward (never the product build): libcvft_stamps.so
set -e
cd "$(dirname "$0")/../cosyvoice_lora_finetune_framework_amd/csrc"
bash build.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -DA32_STAMPS -c attn_mfma32.hip -o build/attn_mfma32_stamps.o
objs=$(ls build/*.o | grep -v attn_mfma32)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcvft_stamps.so $objs build/attn_mfma32_stamps.o
echo "built $(realpath ../libcvft_stamps.so)"
