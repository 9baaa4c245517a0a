#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05q
{
python -m pytest tests/test_gpu_amp_unit.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for v in base amp_lb2 base amp_lb2; do
  if [ $v = base ]; then unset EGREGORA_AMD_LIB; else export EGREGORA_AMD_LIB=variants/lib_$v.so; fi
  echo "== $v"; python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
done
} | tee gpurun_out/r05q/amp_preload.txt
