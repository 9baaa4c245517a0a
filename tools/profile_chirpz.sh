#!/bin/bash
# rocprofv3 evidence for the Fat-Llama path that lengths WITHOUT a packed plan take (paired chirp-z: k_pz_rowconv, k_pzpair,
# k_pzcol; csrc/egr_fatllama_pz.hip), on the two lengths bench.py reports in parts.fatllama_arbitrary_length (60 s + 2 samples:
# even/odd packing; 60 s + 1 sample: channel pairs).  Run through gpurun from the repo root:  tools/profile_chirpz.sh r03
# Counter passes are counters-only (--kernel-trace + --pmc), FETCH_SIZE and WRITE_SIZE in separate passes.
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PROBE_N=2880002,2880001
PROBE_ITERS=800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pz_stats -o pz -- python tools/probe_fatllama_lengths.py > $OUT/pz_stats.log 2>&1
PROBE_ITERS=20 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pz_pmc_fetch -o pz -- python tools/probe_fatllama_lengths.py > $OUT/pz_pmc_fetch.log 2>&1
PROBE_ITERS=20 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pz_pmc_write -o pz -- python tools/probe_fatllama_lengths.py > $OUT/pz_pmc_write.log 2>&1
PROBE_ITERS=20 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pz_pmc_wait -o pz -- python tools/probe_fatllama_lengths.py > $OUT/pz_pmc_wait.log 2>&1
python - "$OUT" <<'PY' > $OUT/pz_summary.txt 2>&1
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
def short(n): return n.split("(")[0].replace("void ", "").replace("egr::", "")
p = glob.glob(f"{out}/pz_stats/**/*kernel_stats.csv", recursive=True)
print("== kernel stats of tools/probe_fatllama_lengths.py on N = 2 880 002 and 2 880 001, stereo, 800 iterations (3 runs each) ==")
for r in list(csv.DictReader(open(p[0])))[:14]:
    print(f"{short(r['Name'])[:70]:70s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} pct={r['Percentage']}")
for d, ctr in (("pz_pmc_fetch", "FETCH_SIZE"), ("pz_pmc_write", "WRITE_SIZE")):
    ps = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(ps[0])):
        if r.get("Counter_Name") == ctr:
            acc[short(r["Kernel_Name"])][0] += float(r["Counter_Value"]); acc[short(r["Kernel_Name"])][1] += 1
    print(f"== {ctr} per launch (KiB as reported -> MB; FETCH also x2-corrected per MI355X_MICROARCH.md) ==")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        mb = v[0] / v[1] * 1024 / 1e6
        print(f"{k[:70]:70s} launches={v[1]:6d} per_launch={mb:9.2f} MB" + (f"  (x2: {2*mb:9.2f} MB)" if ctr == "FETCH_SIZE" else ""))
ps = glob.glob(f"{out}/pz_pmc_wait/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(ps[0])):
    k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
print("== SQ counters, per-launch averages ==")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    n = max(cnt[k], 1); wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{k[:60]:60s} launches={n:6d} wait_any={v.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={v.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
          f"lds_conflict/lds_active={v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_ACTIVE_INST_LDS',1),1):.2f} valu_insts={v.get('SQ_INSTS_VALU',0)/n:.3e}")
PY
cat $OUT/pz_stats.log | grep "^N =" ; cat $OUT/pz_summary.txt
# (run BEFORE tools/profile_round.sh when both share a directory: make_traffic_json.py merges the pz_pmc_* passes) -- raw data is dropped
# by profile_round.sh's clean-up or here when run alone
if [ "${PZ_KEEP_RAW:-0}" != "1" ] && [ "${PZ_BEFORE_ROUND:-0}" != "1" ]; then rm -rf $OUT/pz_stats $OUT/pz_pmc_fetch $OUT/pz_pmc_write $OUT/pz_pmc_wait; fi
