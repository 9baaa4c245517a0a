#!/bin/bash
# same-box A/B of one environment switch inside the whole chain (bench.py --lean), interleaved three times:  tools/r06_same_box_ab.sh VAR [tag]
set -u
export TMPDIR=/tmp
VAR=${1:-EGR_C3_BREG}
TAG=${2:-$VAR}
mkdir -p gpurun_out/r06
for v in 0 1 0 1 0 1; do
  export $VAR=$v
  timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --lean 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v', 'xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'flashsr stage', round(d['parts']['flashsr_stage_ms'],2), 'fatllama stage', round(d['parts']['fatllama_stage_ms'],2))"
done 2>&1 | tee gpurun_out/r06/same_box_ab_$TAG.txt
