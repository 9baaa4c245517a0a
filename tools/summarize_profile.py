"""Summarise rocprofv3 CSV output of tools/profile_round.sh: per-kernel stats + per-launch HBM traffic from the
PMC passes (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x,
see MI355X_MICROARCH.md section HBM -- both raw and corrected figures are printed)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    r = glob.glob(f"{out}/{pattern}", recursive=True)
    return r[0] if r else None


st = find("stats/**/*kernel_stats.csv")
if st:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(st) as f:
        rows = list(csv.DictReader(f))
    for r in rows[:16]:
        print("%-86s calls=%6s avg_us=%10.1f total_ms=%9.1f pct=%5.1f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                       float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    p = find(f"{tag}/**/*counter_collection.csv")
    if not p:
        print(f"no {ctr} csv")
        continue
    acc = defaultdict(lambda: [0.0, 0])
    with open(p) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != ctr:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    print(f"== {ctr} per launch (KiB as reported; bytes; x2-corrected bytes for FETCH) ==")
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:10]:
        per = v / max(n, 1)
        print(f"{k[:70]:70s} launches={n:5d} per_launch={per:12.1f} KiB = {per*1024/1e6:9.2f} MB"
              + (f"  (x2: {2*per*1024/1e6:9.2f} MB)" if ctr == "FETCH_SIZE" else ""))
