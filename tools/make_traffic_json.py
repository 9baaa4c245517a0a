#!/usr/bin/env python3
"""profiles/traffic.json from the counters-only rocprofv3 passes of tools/profile_round.sh:
per kernel instantiation, HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB; the x2 is the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md, section HBM).  bench.py reads it for `roofline.traffic`.
  python tools/make_traffic_json.py gpurun_out/r02 profiles/r02"""
import csv, glob, json, subprocess, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
build = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
per = {}
# chain passes (tools/profile_round.sh) first, then the chirp-z passes of tools/profile_chirpz.sh when they sit in the same directory:
# a kernel keeps the numbers of the FIRST pass that saw it (the chain's shapes are the ones bench.py's roofline objects time)
for ctr, dirs in (("FETCH_SIZE", ("pmc_fetch", "pz_pmc_fetch")), ("WRITE_SIZE", ("pmc_write", "pz_pmc_write"))):
    per[ctr] = {}
    for d in dirs:
        ps = glob.glob(f"{src}/{d}/**/*counter_collection.csv", recursive=True)
        if not ps:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(ps[0])):
            if r.get("Counter_Name") == ctr:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("egr::", "")
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, v in acc.items():
            per[ctr].setdefault(k, v[0] / max(v[1], 1) * 1024.0)
def csrc_sha():
    import hashlib, pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    h = hashlib.sha256()
    d = root / "comfyui-egregora-audio-super-resolution_amd" / "csrc"
    files = sorted([f for f in d.iterdir() if f.suffix in (".hip", ".h", ".cpp", ".inc") or f.name == "Makefile"]) + [root / "include" / "egregora_amd.h"]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


out = {"_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from the rocprofv3 PMC passes (separate runs; FETCH x2 gfx950 correction per "
                f"MI355X_MICROARCH.md section HBM); source: {tag}/chain60_summary.txt; Fat-Llama loop kernels are per ONE-channel launch "
                "(two channel pipelines run concurrently)",
       "source": f"{tag} ({src}), library built from commit {build}",
       "csrc_sha": csrc_sha(),          # bench.py compares it with the sources it runs on (same function)
       "kernels": {}}
for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
    f, w = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
    if 2 * f + w >= 1e6 and k.startswith("k_"):
        out["kernels"][k] = {"bytes": round(2 * f + w), "fetch_x2": round(2 * f), "write": round(w)}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v["bytes"] for k, v in list(out["kernels"].items())[:40]}, indent=0))
