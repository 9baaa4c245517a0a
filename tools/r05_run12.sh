#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu_s.txt 2>&1
tail -3 $OUT/pytest_gpu_s.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
for v in base bm128 base bm128; do
  if [ $v = base ]; then unset EGREGORA_AMD_LIB; else export EGREGORA_AMD_LIB=variants/lib_$v.so; fi
  python bench.py --only flashsr --lean --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), {k:(v['launches'], round(v['ms'],2)) for k,v in d['parts']['conv_variants'].items() if '256, 128' in k or '128, 128' in k})"
done 2>&1 | tee gpurun_out/r05/bm128_ab.txt
