#!/bin/bash
# dev: sweep Fat-Llama loop-kernel knobs on the C3 shape (run through gpurun)
run() { echo -n "$* : "; env "$@" python tools/bench_fatllama_only.py; }
run EGR_FL_M1=0
for m1 in 450 500 576 600 625 640 720 750 768 800 900 960 1000; do run EGR_FL_M1=$m1; done
run EGR_FL_TC=4
