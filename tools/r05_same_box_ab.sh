#!/bin/bash
# same-box A/B: this round's FlashSR kernels on / off (pipelined 3x3 kernel, fused AMP unit), interleaved
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for v in r04like r05 r04like r05 r04like r05; do
  if [ $v = r04like ]; then export EGR_S3_CONV3X3=1 EGREGORA_FLASHSR_FUSED_AMP=0; else unset EGR_S3_CONV3X3 EGREGORA_FLASHSR_FUSED_AMP; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lean 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'flashsr stage', round(d['parts']['flashsr_stage_ms'],2))"
done 2>&1 | tee gpurun_out/r05/same_box_ab.txt
