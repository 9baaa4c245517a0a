#!/bin/bash
# dev: A/B of Fat-Llama library variants on ONE box: tools/probe_fl_ab.sh variants/lib_a.so - -@EGR_FL_COL_TC=8 ...
# ("-" = the shipped library; "@VAR=VALUE" adds an environment variable).  PROBE_N lengths through tools/probe_fatllama_lengths.py,
# three rounds interleaved so that box drift shows.
cd "$(dirname "$0")/.."
export PROBE_N=${PROBE_N:-2880000}
for round in 1 2 3; do
  for spec in "$@"; do
    lib=${spec%%@*}; kv=""; [ "$spec" != "$lib" ] && kv=${spec#*@}
    echo "== round $round $spec"
    ( if [ "$lib" != "-" ]; then export EGREGORA_AMD_LIB="$PWD/$lib"; fi
      [ -n "$kv" ] && export "$kv"
      python tools/probe_fatllama_lengths.py 2>&1 | grep "^N =" | cut -c1-60 )
  done
done
