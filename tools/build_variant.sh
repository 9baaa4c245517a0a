#!/bin/bash
# dev: build variants/lib_<name>.so = the shipped objects with ONE source recompiled under extra -D flags
#   tools/build_variant.sh <name> <source in csrc> "<flags>"        then  EGREGORA_AMD_LIB=variants/lib_<name>.so python ...
set -e
cd "$(dirname "$0")/../comfyui-egregora-audio-super-resolution_amd/csrc"
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p ../../variants build
NOPK=""
case $SRC in egr_fatllama*|egr_glue*|egr_nn_ops*|egr_nn_amp*|egr_nn_wino4*|egr_flashsr_pack*) NOPK="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -DEGR_RADIX_8_9 $NOPK $FLAGS -c $SRC -o build/var_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/var_" | grep -v "build/canary_" | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/var_$NAME.o -o ../../variants/lib_$NAME.so
echo built variants/lib_$NAME.so
