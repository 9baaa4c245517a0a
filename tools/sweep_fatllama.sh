#!/bin/bash
# dev: sweep Fat-Llama loop-kernel knobs on the C3 shape (run through gpurun)
run() { echo -n "$* : "; env "$@" EGR_FL_TWPOW=1 python tools/bench_fatllama_only.py; }
for t in 512 1024; do run EGR_FL_THREADS=$t; done
for m1 in 576 600 640 720 750 768 900 960 1000; do run EGR_FL_THREADS=512 EGR_FL_M1=$m1; done
run EGR_FL_THREADS=512 EGR_FL_TC=4
run EGR_FL_THREADS=512 EGR_FL_TC=16
