#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# Writes gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}/... and a text summary; copy what should be judged to profiles/.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o chain -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > $OUT/bench_stats.log 2>&1
grep '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench_stats.json
# PMC passes: counters only, each in its own run (no trace domains besides kernel-trace); short Fat-Llama loop
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o chain -- python bench.py --steps 1 --warmup 0 --iters 20 --no-cpu-baseline --lean > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o chain -- python bench.py --steps 1 --warmup 0 --iters 20 --no-cpu-baseline --lean > $OUT/bench_pmc_write.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
