#!/bin/bash
# round 5, GPU call 1: allocation costs, the full -m gpu suite with its printed figures, the bench line with the first-call trace,
# the round's rocprofv3 passes (chain + chirp-z).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
tools/ubench/malloc_cost > $OUT/malloc_cost.log 2>&1
EGR_FSR_TRACE=1 python bench.py > $OUT/bench_default.log 2>&1
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json
python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu_s.txt 2>&1
tail -3 $OUT/pytest_gpu_s.txt
tools/profile_round.sh r05a > $OUT/profile_round.log 2>&1
tools/profile_chirpz.sh r05a > $OUT/profile_chirpz.log 2>&1
tail -30 $OUT/counters.txt
cat $OUT/malloc_cost.log
grep "egr_flashsr_infer\]" $OUT/bench_default.log | head
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05a/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_arbitrary_length")}, d["parts"]["flashsr_stage_ms"], d["parts"]["fatllama_stage_ms"], d["parts"]["flashsr_stage_first_call_ms"], d["roofline"]["frac"])
PY
