#!/bin/bash
# dev: PMC counters of the Fat-Llama loop kernels on the C3 shape
export TMPDIR=/tmp
cat > /tmp/fl_small.py <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe
x=torch.from_numpy((0.3*np.random.default_rng(0).standard_normal((2,2880000))).astype(np.float32)).cuda()
fe.enhance_device(x,1,40,0.6,True,False,True,True); torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_fl1 -o f -- python /tmp/fl_small.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/pmc_fl2 -o f -- python /tmp/fl_small.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
from collections import defaultdict
for d in ("pmc_fl1","pmc_fl2"):
    ps=glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv",recursive=True)
    if not ps: print("no csv",d); continue
    acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(ps[0])):
        k=r["Kernel_Name"].split("(")[0][-24:]
        if "k_row" in k or "k_col<1>" in k:
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in acc.items():
        print(k,{a:"%.3e"%(b/cnt[k][a]) for a,b in v.items()})
PY
