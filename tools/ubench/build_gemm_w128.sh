#!/bin/bash
# builds tools/ubench/gemm_w128 (extra flags: e.g. -DSGB=0); prints the kernel's register / spill figures
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w "$@" -I include -I comfyui-egregora-audio-super-resolution_amd/csrc tools/ubench/gemm_w128.hip \
  -L comfyui-egregora-audio-super-resolution_amd -legregora_amd -o tools/ubench/gemm_w128 -save-temps=obj 2>&1 | grep -E "error" | head
grep -E "; NumVgprs|; NumAgprs|; ScratchSize|; Occupancy" tools/ubench/gemm_w128-hip-amdgcn-amd-amdhsa-gfx950.s | tr '\n' ' '; echo
