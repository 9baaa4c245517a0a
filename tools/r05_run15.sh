#!/bin/bash
set -u
export TMPDIR=/tmp
TAG=r05
OUT=gpurun_out/$TAG
mkdir -p $OUT
PZ_BEFORE_ROUND=1 tools/profile_chirpz.sh $TAG > $OUT/profile_chirpz.log 2>&1
tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
head -12 $OUT/summary.txt | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.log 2>&1
grep '^{' $OUT/bench_driver_flags.log | tail -1 > $OUT/chain60_bench_driver_flags_20steps.json
python -c "
import json; d=json.load(open('$OUT/chain60_bench_driver_flags_20steps.json')); print('20 steps:', d['value'], d['ms_per_step'], d['value_arbitrary_length'], d['parts']['flashsr_stage_ms'], d['parts']['fatllama_stage_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_fatllama_chirpz']['kernel'], d['roofline_fatllama_chirpz']['traffic'], d['roofline_fatllama_chirpz']['frac'])"
