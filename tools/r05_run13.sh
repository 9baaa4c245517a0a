#!/bin/bash
set -u
export TMPDIR=/tmp
tools/r05_evidence.sh r05
python __graft_entry__.py smoke 2>&1 | tail -1 > gpurun_out/r05/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_flags.log 2>&1
grep '^{' gpurun_out/r05/bench_driver_flags.log | tail -1 > gpurun_out/r05/chain60_bench_driver_flags_20steps.json
python -c "
import json; d=json.load(open('gpurun_out/r05/chain60_bench_driver_flags_20steps.json')); print('20 steps:', d['value'], d['ms_per_step'], d['value_arbitrary_length'], d['parts']['flashsr_stage_ms'], d['parts']['fatllama_stage_ms'], d['roofline']['frac'], d['cpu_baseline']['value'] if 'cpu_baseline' in d else None)"
