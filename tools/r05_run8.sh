#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05h
mkdir -p $OUT
EGR_FSR_PROFILE_DUMP=1 python bench.py --only flashsr --lean --steps 2 --warmup 1 --no-cpu-baseline > $OUT/profile_dump.log 2>&1
grep "egr_flashsr profile" $OUT/profile_dump.log | sort -k4 -n -r | head -70
python -m pytest tests/test_gpu_fatllama.py -m gpu -x -q -s -p no:cacheprovider -k "real_gating" > $OUT/pytest_gating.txt 2>&1; tail -12 $OUT/pytest_gating.txt
python -m pytest tests/test_devices_partition.py tests/test_gpu_split_h2.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_new.txt 2>&1; tail -3 $OUT/pytest_new.txt
PROBE_OUT=$OUT/oracle32_c3_plus2.json python tools/probe_c3_plus2_oracle.py > $OUT/probe_oracle.log 2>&1; tail -3 $OUT/probe_oracle.log
