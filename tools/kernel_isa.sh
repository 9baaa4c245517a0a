#!/bin/bash
# Disassembly of one shipped kernel (code objects extracted under /tmp): tools/kernel_isa.sh '<mangled-name substring>' > out.s
# and its instruction-class picture (M mfma, r / w LDS read / write, G / S global load / store, | s_waitcnt, B barrier, v VALU, s SALU, J branch)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=${LIB:-$ROOT/comfyui-egregora-audio-super-resolution_amd/libegregora_amd.so}
T=$(mktemp -d /tmp/egr_isa.XXXXXX)
cp "$LIB" "$T/lib.so"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null)
for o in "$T"/*amdgcn*gfx950; do
  [ -s "$o" ] || continue
  /opt/rocm/lib/llvm/bin/llvm-objdump -d "$o" | awk -v pat="$1" '
    /^[0-9a-f]+ <.*>:$/ { on = index($0, pat) > 0 }
    on { print }'
done
rm -rf "$T"
