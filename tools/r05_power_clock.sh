#!/bin/bash
# shader clock and socket power under the dominant GEMM, random vs all-zero operands (same instruction stream): is the kernel power-limited?
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05u
{
for z in 0 1; do
  if [ $z = 1 ]; then export S3_ZERO=1; else unset S3_ZERO; fi
  ( for i in $(seq 1 70); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*Power (W): \([0-9.]*\)/\1 W/' | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/r05u/smi_zero$z.txt &
  echo -n "zero=$z, 6000 launches: "; S3_REPS=12000 S3_SCH=1 S3_BN=256 S3_BM=128 S3_XCD=1 tools/ubench/conv_s3_base 936 32 16 512 512 1 | sed 's/cs=[^ ]* //'
  wait
  echo "rocm-smi samples next to it (zero=$z):"; cat gpurun_out/r05u/smi_zero$z.txt | tr '\n' ';' | cut -c1-1500; echo
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05u/power_clock.txt
