#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes written by tools/profile_round.sh (pmc_mfma, pmc_wait)."""
import csv, glob, sys
from collections import defaultdict

out = sys.argv[1]
for d, keep in (("pmc_mfma", ("k_conv_s3", "k_conv1d_s3", "k_conv3x3_is", "k_conv1d_rb", "k_bgemm_s3", "k_wino4", "k_conv_igemm", "k_snake")), ("pmc_wait", ("k_row", "k_col", "k_pz"))):
    ps = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not ps:
        print("no counter csv under", d)
        continue
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(ps[0])):
        k = r["Kernel_Name"].split("(")[0]
        k = k[k.find("k_"):] if "k_" in k else k
        if any(t in k for t in keep):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    print(f"== {d}: per-launch averages (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES count cycles)")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
        n = max(cnt[k].values())
        avg = {a: b / cnt[k][a] for a, b in v.items()}
        line = f"{k[:60]:60s} launches {n:5d} " + " ".join(f"{a}={b:.3e}" for a, b in sorted(avg.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
            # MFMA busy cycles are summed over the 4 SIMDs of a CU: busy fraction of the matrix pipes = MFMA_BUSY / (4 * BUSY_CU)
            line += f" | mfma_busy_frac={avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * avg['SQ_BUSY_CU_CYCLES']):.3f}"
        if "SQ_WAIT_ANY" in avg and avg.get("SQ_WAVE_CYCLES"):
            line += f" | wait_any={avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES']:.2f} wait_inst={avg.get('SQ_WAIT_INST_ANY', 0) / avg['SQ_WAVE_CYCLES']:.2f} active={avg.get('SQ_ACTIVE_INST_ANY', 0) / avg['SQ_WAVE_CYCLES']:.2f}"
        if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_ACTIVE_INST_LDS"):
            line += f" lds_conflict/lds_active={avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_ACTIVE_INST_LDS']:.2f}"
        print(line)
