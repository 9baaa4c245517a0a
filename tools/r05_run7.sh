#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
{
for shape in "936 32 16 512 512 1" "936 16 8 1024 1024 1" "26 128 64 256 512 3"; do
  for tag in base sgb2 sgb3 base sgb2 sgb3; do
    echo -n "$tag: "; S3_SCH=1 S3_BN=256 S3_BM=128 S3_XCD=1 tools/ubench/conv_s3_$tag $shape | sed 's/cs=[^ ]* //'
  done
done
EGR_FSR_TRACE=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
grep "egr_flashsr" $OUT/bench.log | head -24
grep '^{' $OUT/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['parts']; print({k:d[k] for k in ('value','ms_per_step','value_arbitrary_length')}, {k:p[k] for k in ('flashsr_stage_ms','fatllama_stage_ms','flashsr_stage_first_call_ms','torch_runtime_warmup_ms','flashsr_handle_build_ms','node_boundary_ms')})"
} 2>&1 | grep -v amdgpu.ids | tee $OUT/sgb_and_bench.txt
