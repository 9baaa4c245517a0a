#!/bin/bash
# dev: per-kernel durations of the paired chirp-z loop for a list of "L,nc" splits (one channel pipeline), rocprofv3 stats
#   tools/pz_sweep.sh <length> <split> [<split> ...]       (run through gpurun from the repo root)
export TMPDIR=/tmp
N=$1; shift
for sp in "$@"; do
  rm -rf /tmp/pzs
  lib=""
  case $sp in lib:*) lib=variants/lib_${sp#lib:}.so; sp=${PZ_SPLIT:-};; esac
  echo "== split=$sp lib=$lib"
  EGREGORA_AMD_LIB=$lib EGR_PZ_SPLIT=$sp EGR_FL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pzs -o t -- python tools/probe_pz_time.py 200 $N > /tmp/pzs.log 2>&1
  grep "^n=" /tmp/pzs.log | tail -1
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pzs/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows:
    if 'k_pz' in r['Name'] and int(r['Calls']) > 100:
        print("   %-70s calls %6s avg %8.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
