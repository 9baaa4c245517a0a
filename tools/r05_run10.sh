#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
{
for tl in 256 240 496; do EGR_AMP_TL=$tl python tools/bench_amp_unit.py; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/amp_unit_tl.txt
