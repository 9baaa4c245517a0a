#!/bin/bash
# round 5, GPU call 2: the pipelined input-stationary 3x3 kernel -- parity tests, A/B per launch and per stage; first-call stage trace
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
python -m pytest tests/test_gpu_split_h2.py -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_split_h2.txt 2>&1; tail -3 $OUT/pytest_split_h2.txt
for v in 1 2 1 2; do EGR_S3_CONV3X3=$v python tools/bench_conv3x3_is.py; done 2>&1 | tee $OUT/conv3x3_ab.txt
for v in 1 2; do EGR_S3_CONV3X3=$v python bench.py --only flashsr --lean --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_flashsr_v$v.log 2>&1; grep '^{' $OUT/bench_flashsr_v$v.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('EGR_S3_CONV3X3=$v', d['ms_per_step'], d['parts']['flashsr_stage_ms'], {k:(v['launches'], round(v['ms'],2)) for k,v in d['parts']['conv_variants'].items() if v['ms']>2})"; done 2>&1 | tee $OUT/stage_ab.txt
python -m pytest tests/test_gpu_flashsr.py -m gpu -x -q -s -p no:cacheprovider -k "full_size or bench_shape or different_level" > $OUT/pytest_flashsr_full.txt 2>&1; tail -3 $OUT/pytest_flashsr_full.txt
EGR_FSR_TRACE=2 EGREGORA_FLASHSR_STREAMS=1 python bench.py --only flashsr --lean --steps 1 --warmup 0 --no-cpu-baseline > $OUT/first_call_trace.log 2>&1; grep "egr_flashsr" $OUT/first_call_trace.log | head -30
