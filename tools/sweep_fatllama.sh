#!/bin/bash
# dev: sweep Fat-Llama loop-kernel knobs on the C3 shape (run through gpurun)
run() { echo -n "$* : "; env "$@" python tools/bench_fatllama_only.py 2>&1 | grep -v amdgpu | tail -1; }
run EGR_FL_M1=0
for m1 in 360 400 450 480 500 576 600 640 720 750 768 800 900 960 1000; do run EGR_FL_M1=$m1; done
for tc in 4 16; do run EGR_FL_TC=$tc; done
for th in 256 1024; do run EGR_FL_THREADS=$th; done
run EGR_FL_STREAMS=1
