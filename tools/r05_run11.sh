#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
python -m pytest tests/test_gpu_amp_unit.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
python tools/bench_amp_unit.py 2>&1 | grep -v amdgpu.ids | tee $OUT/amp_unit.txt
tools/r05_evidence.sh r05 
