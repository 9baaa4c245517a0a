#!/bin/bash
# round 6: joules per TFLOP of the two f16 MFMA shapes and of the shipped GEMM, from rocm-smi samples next to multi-second runs
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06u
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_energy tools/ubench/mfma_f16_energy.hip || exit 1
sample() { ( for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*(\([0-9]*\)Mhz).*/\1/; s/.*Power (W): \([0-9.]*\)/\1/' | tr '\n' ' '; echo; sleep 0.25; done ) > $2; }
summ() { awk 'NR > 4 && NF >= 2 { c += $1; w += $2; n++ } END { if (n) printf("  rocm-smi mean over %d samples (first second dropped): %.0f MHz, %.0f W\n", n, c / n, w / n) }' $1; }
{
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  sample 20 gpurun_out/r06u/smi.txt &
  /tmp/mfma_f16_energy $cfg 5
  wait; summ gpurun_out/r06u/smi.txt
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06u/mfma_energy.txt
