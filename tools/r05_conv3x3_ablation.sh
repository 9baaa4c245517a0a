#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
{
for v in base noepi nob nohalo nomma onlymma loadonly base; do
  if [ $v = base ]; then unset EGREGORA_AMD_LIB; else export EGREGORA_AMD_LIB=variants/lib_c3_$v.so; fi
  echo -n "$v: "; EGR_S3_CONV3X3=1 REPS=60 python tools/bench_conv3x3_is.py
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/c3_ablation.txt
