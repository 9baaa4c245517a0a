#!/bin/bash
# VGPR / spill / LDS figures of the shipped kernels (code objects are extracted under /tmp, never next to the library).
# usage: tools/kernel_resources.sh [regex]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=${LIB:-$ROOT/comfyui-egregora-audio-super-resolution_amd/libegregora_amd.so}
T=$(mktemp -d /tmp/egr_res.XXXXXX)
cp "$LIB" "$T/lib.so"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null)
for o in "$T"/*amdgcn*gfx950; do
  [ -s "$o" ] || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$o" | python3 -c '
import re, sys
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
cur = {}
def flush():
    if cur.get("name") and (pat is None or pat.search(cur["name"])):
        print("%-110s vgpr %3s agpr %3s spill %3s lds %6s scratch %5s" % (cur["name"][:110], cur.get("vgpr","?"), cur.get("agpr","?"), cur.get("spill","?"), cur.get("lds","?"), cur.get("scratch","?")))
for ln in sys.stdin:
    ln = ln.strip()
    m = re.match(r"- \.agpr_count:\s+(\d+)", ln)
    if m: flush(); cur.clear(); cur["agpr"] = m.group(1); continue
    for k, key in ((".name:", "name"), (".vgpr_count:", "vgpr"), (".vgpr_spill_count:", "spill"), (".group_segment_fixed_size:", "lds"), (".private_segment_fixed_size:", "scratch"), (".agpr_count:", "agpr")):
        if ln.startswith(k) or ln.startswith("- " + k):
            cur[key] = ln.split(":", 1)[1].strip()
flush()
' "$@" | c++filt
done
rm -rf "$T"
