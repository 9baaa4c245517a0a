#!/bin/bash
# round 5 evidence run: the whole -m gpu suite with its printed figures, the rocprofv3 passes (chirp-z first, then the chain), the bench line
set -u
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $OUT/pytest_gpu_s.txt 2>&1
tail -3 $OUT/pytest_gpu_s.txt
PZ_BEFORE_ROUND=1 tools/profile_chirpz.sh $TAG > $OUT/profile_chirpz.log 2>&1
tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
tail -40 $OUT/counters.txt | cut -c1-400
cat $OUT/pz_summary.txt | head -40
python bench.py > $OUT/bench_default.log 2>&1
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/chain60_bench_driver_flags.json
python - <<PY
import json
d=json.load(open("$OUT/chain60_bench_driver_flags.json"))
p=d["parts"]
print({k:d[k] for k in ("value","ms_per_step","value_arbitrary_length")}, {k:p[k] for k in ("flashsr_stage_ms","fatllama_stage_ms","flashsr_stage_first_call_ms","node_boundary_ms")}, d["roofline"]["frac"], d["roofline"]["traffic"], d.get("roofline_fatllama_chirpz",{}).get("traffic"))
PY
du -sh $OUT
