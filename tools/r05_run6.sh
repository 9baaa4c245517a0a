#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
{
for v in 1 2 1 2; do EGR_S3_CONV3X3=$v REPS=60 python tools/bench_conv3x3_is.py; done
python -m pytest tests/test_gpu_split_h2.py -m gpu -x -q -p no:cacheprovider -k "input_stationary" 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee $OUT/c3_ab2.txt
