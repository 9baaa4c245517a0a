for cfg in "625 8" "500 16" "500 8" "450 16" "480 8" "600 8" "750 8" "900 8" "360 16" "250 16"; do set -- $cfg; for st in 2 1; do
EGR_FL_M1=$1 EGR_FL_TC=$2 EGR_FL_STREAMS=$st timeout 300 python bench.py --only fatllama --steps 3 --warmup 1 --no-cpu-baseline --lean 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_fatllama']; print('M1=$1 TC=$2 streams=$st', round(d['ms_per_step'],2), d['config']['fatllama_split'], round(r['k_row_ms']*1e3,1), round(r['k_col_ms']*1e3,1))" 2>&1 | tail -1
done; done
