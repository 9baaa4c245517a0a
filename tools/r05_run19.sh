#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05s
{
python -m pytest tests/test_gpu_amp_unit.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
AMP_C=32 AMP_L=122880 python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | sed 's/; fused vs four.*//'
} | tee gpurun_out/r05s/amp_interior.txt
