#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05r
{
for v in base amp_t224l4 amp_t240l4 amp_t224l3 base amp_t224l4; do
  if [ $v = base ]; then unset EGREGORA_AMD_LIB; else export EGREGORA_AMD_LIB=variants/lib_$v.so; fi
  echo "== $v"; python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
done
} | tee gpurun_out/r05r/amp_occupancy.txt
