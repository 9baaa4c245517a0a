#!/bin/bash
# round 5: the fused AMP unit -- operator tests, stage A/B (four launches per unit vs one), full-size float64 gates with it
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
python -m pytest tests/test_gpu_amp_unit.py -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_amp.txt 2>&1; tail -30 $OUT/pytest_amp.txt | cut -c1-200
for v in 0 16 32 0 32; do
  if [ $v = 0 ]; then export EGREGORA_FLASHSR_FUSED_AMP=0; else export EGREGORA_FLASHSR_FUSED_AMP=1 EGREGORA_FLASHSR_FUSED_AMP_MAX_C=$v; fi
  EGR_FSR_PROFILE_DUMP=1 python bench.py --only flashsr --lean --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_amp_$v.log 2>&1
  grep '^{' $OUT/bench_amp_$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fused amp max C = $v:', round(d['ms_per_step'],2), round(d['parts']['flashsr_stage_ms'],2), {k:(v['launches'], round(v['ms'],2)) for k,v in d['parts']['conv_variants'].items() if 'conv1d' in k or 'amp' in k})"
done 2>&1 | tee $OUT/amp_stage_ab.txt
unset EGREGORA_FLASHSR_FUSED_AMP EGREGORA_FLASHSR_FUSED_AMP_MAX_C
grep "k_amp_unit" $OUT/bench_amp_32.log | tail -36 | cut -c1-150
python -m pytest tests/test_gpu_flashsr.py -m gpu -x -q -s -p no:cacheprovider -k "full_size or bench_shape or different_level" > $OUT/pytest_flashsr_full.txt 2>&1; tail -4 $OUT/pytest_flashsr_full.txt | cut -c1-300
