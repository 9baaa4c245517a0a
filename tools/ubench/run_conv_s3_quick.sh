#!/bin/bash
# base / NOEPI / NOSTORE timings of the fp16-scheme k_conv_s3 on the dominant GEMM shapes (see run_conv_s3_ablation.sh)
set -e
cd "$(dirname "$0")/../.."
INC="-I comfyui-egregora-audio-super-resolution_amd/csrc -I include"
for tag in base; do
  D=""; [ $tag != base ] && D="-DS3_ABL_$tag"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEGR_RADIX_8_9 -w $INC $D tools/ubench/conv_s3_harness.hip -o /tmp/conv_s3_$tag &
done
wait
for shape in "936 32 16 512 512 1" "936 16 8 1024 1024 1" "26 128 64 256 512 3"; do
  for tag in base; do
    echo -n "sch=1 $tag: "; S3_SCH=1 S3_BN=256 S3_BM=128 /tmp/conv_s3_$tag $shape | sed 's/cs=[^ ]* //'
  done
done
