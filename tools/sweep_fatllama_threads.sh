for cfg in "512 8" "256 8" "1024 8" "512 4" "256 4" "512 16"; do set -- $cfg
EGR_FL_THREADS=$1 EGR_FL_TC=$2 timeout 300 python bench.py --only fatllama --steps 3 --warmup 1 --no-cpu-baseline --lean 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_fatllama']; print('threads=$1 TC=$2', round(d['ms_per_step'],2), round(r['k_row_ms']*1e3,1), round(r['k_col_ms']*1e3,1))" 2>&1 | tail -1
done
EGR_FL_SCHED=0 python bench.py --only fatllama --steps 3 --warmup 1 --no-cpu-baseline --lean 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_fatllama']; print('SCHED=0', round(d['ms_per_step'],2), round(r['k_row_ms']*1e3,1), round(r['k_col_ms']*1e3,1))"
python tools/trace_fatllama_phases.py 1 2>&1 | grep -v amdgpu
