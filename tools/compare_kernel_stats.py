"""Dev: per-kernel totals of two rocprofv3 kernel_stats.csv files side by side (per forward when --div is given).
usage: python tools/compare_kernel_stats.py NEW.csv OLD.csv [new_div] [old_div]"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        k = re.sub(r'\(.*', '', r['Name']).replace('egr::', '').replace('void ', '')
        c, t = d.get(k, (0, 0.0))
        d[k] = (c + int(r['Calls']), t + float(r['TotalDurationNs']) / 1e6)
    return d
new, old = load(sys.argv[1]), load(sys.argv[2])
nd = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
od = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
print("total per forward: new %.1f ms, old %.1f ms" % (sum(v[1] for v in new.values()) / nd, sum(v[1] for v in old.values()) / od))
for k in sorted(set(new) | set(old), key=lambda k: -max(new.get(k, (0, 0))[1] / nd, old.get(k, (0, 0))[1] / od))[:45]:
    n, o = new.get(k, (0, 0.0)), old.get(k, (0, 0.0))
    print("%-48s new n=%6.1f %8.2f ms | old n=%6.1f %8.2f ms | %+7.2f" % (k[:48], n[0] / nd, n[1] / nd, o[0] / od, o[1] / od, n[1] / nd - o[1] / od))
