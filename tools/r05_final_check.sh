#!/bin/bash
# the round's closing run on the final build: the whole -m gpu suite with its printed figures, smoke, the bench line with the driver's flags
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05final
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu_s.txt 2>&1
tail -2 $OUT/pytest_gpu_s.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/chain60_bench_driver_flags.json
python -c "
import json; d=json.load(open('$OUT/chain60_bench_driver_flags.json')); p=d['parts']; print(d['value'], d['ms_per_step'], d['value_arbitrary_length'], p['flashsr_stage_ms'], p['fatllama_stage_ms'], p['flashsr_stage_first_call_ms'], d['roofline']['frac'], d['roofline']['kernel'], {k:round(v['ms'],2) for k,v in p['conv_variants'].items() if v['ms']>3})"
