#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
{
EGR_S3_CONV3X3=1 REPS=50 python tools/bench_conv3x3_is.py
EGREGORA_AMD_LIB=variants/lib_c3timing.so EGR_S3_CONV3X3=1 REPS=50 python tools/bench_conv3x3_is.py
EGR_S3_CONV3X3=2 REPS=50 python tools/bench_conv3x3_is.py
} 2>&1 | grep -v amdgpu.ids | tee $OUT/c3_timing.txt
