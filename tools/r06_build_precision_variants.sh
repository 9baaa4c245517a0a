#!/bin/bash
# dev (round 6): libraries with the precision switches of csrc/egr_fatllama_wl.h (EGR_WL_HILO bits) and egr_fft_device.h (EGR_BFLY_HILO)
# usage: tools/r06_build_precision_variants.sh "name:flags" ...
set -e
cd "$(dirname "$0")/.."
build() { tools/build_variant.sh "$1" egr_fatllama.hip "$2" > /dev/null; echo "built $1 ($2)"; }
i=0
for spec in "$@"; do
  build "${spec%%:*}" "${spec#*:}" &
  i=$((i+1)); if [ $((i % 4)) -eq 0 ]; then wait; fi
done
wait
