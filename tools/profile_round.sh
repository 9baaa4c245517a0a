#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# Writes gpurun_out/<tag>/{stats,pmc_*}/... and text summaries; copy what should be judged to profiles/<tag>/.
# Counter passes are counters-only (--kernel-trace + --pmc, nothing else), one pass per counter group (SQ: 8 slots, TCC: FETCH_SIZE
# and WRITE_SIZE cannot share a pass) as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
# no throw-away warm-up pass under the profiler: every FlashSR launch of the capture then has the bench's 26-row (or 13-row) shape
export EGREGORA_FLASHSR_WARMUP=0
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o chain -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > $OUT/bench_stats.log 2>&1
grep '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench_stats.json
# the same with ONE row group per pass: every contraction launch then has the 26-row shape bench.py's roofline object times
# (egr_flashsr_set_profiling runs one group), so the per-kernel averages of this pass are the ones that must agree with it
EGREGORA_FLASHSR_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_g1 -o chain -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > $OUT/bench_stats_g1.log 2>&1
grep '^{' $OUT/bench_stats_g1.log | tail -1 > $OUT/bench_stats_g1.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o chain -- python bench.py --steps 1 --warmup 1 --iters 20 --no-cpu-baseline --lean > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o chain -- python bench.py --steps 1 --warmup 1 --iters 20 --no-cpu-baseline --lean > $OUT/bench_pmc_write.log 2>&1
# matrix-pipe occupancy of the contraction kernels (north_star: "MFMA-busy counters") and the wait / LDS picture of the Fat-Llama loop
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma -o chain -- python bench.py --steps 1 --warmup 1 --iters 20 --no-cpu-baseline --lean > $OUT/bench_pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_wait -o chain -- python bench.py --only fatllama --steps 1 --warmup 0 --iters 60 --no-cpu-baseline --lean > $OUT/bench_pmc_wait.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python tools/summarize_counters.py $OUT > $OUT/counters.txt 2>&1
python tools/make_traffic_json.py $OUT profiles/$TAG "${BUILD_COMMIT:-unknown}" > $OUT/traffic_kernels.txt 2>&1 && cp profiles/traffic.json $OUT/traffic.json
# gpurun copies back at most 64 MiB: keep the per-kernel statistics, drop the raw traces and counter dumps
for d in stats stats_g1; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${d}_kernel_stats.csv; done
rm -rf $OUT/stats $OUT/stats_g1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_wait $OUT/pz_stats $OUT/pz_pmc_fetch $OUT/pz_pmc_write $OUT/pz_pmc_wait
cat $OUT/summary.txt $OUT/counters.txt
