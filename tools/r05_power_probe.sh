#!/bin/bash
# round 5, GPU call 3: is the matrix pipe power-limited?  random vs zero operands on the shipped contraction kernels, the register-only
# f16 MFMA ceiling, and the shader clock under each load (rocm-smi sampled next to a long run)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
{
tools/ubench/mfma_f16_peak
for shape in "936 32 16 512 512 1" "936 16 8 1024 1024 1"; do
  for z in 0 1; do
    if [ $z = 1 ]; then export S3_ZERO=1; else unset S3_ZERO; fi
    echo -n "zero=$z: "; S3_SCH=1 S3_BN=256 S3_BM=128 S3_XCD=1 tools/ubench/conv_s3_base $shape | sed 's/cs=[^ ]* //'
  done
done
unset S3_ZERO
python tools/bench_conv3x3_is.py
ZERO=1 python tools/bench_conv3x3_is.py
( for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2; sleep 0.5; done ) > $OUT/clocks_conv3x3.txt &
REPS=1500 python tools/bench_conv3x3_is.py
wait
cat $OUT/clocks_conv3x3.txt | sort | uniq -c
rocm-smi --showpower --showclocks 2>/dev/null | head -30
} 2>&1 | grep -v amdgpu.ids | tee $OUT/power_probe.txt
