#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05p
{
for v in base nosin nosnake nomma nothing base; do
  if [ $v = base ]; then unset EGREGORA_AMD_LIB; else export EGREGORA_AMD_LIB=variants/lib_amp_$v.so; fi
  echo "== $v"; python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
done
} | tee gpurun_out/r05p/amp_ablation.txt
