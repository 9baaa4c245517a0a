"""dev: per-phase wall-clock stamps of one k_row / k_col<1> launch on the C3 plan (egr_fatllama_trace_once, csrc/egr_fatllama.hip)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, native
L = native.lib()
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
plan = fe._plan(2880000, ch, 1, 0)
x = (3000 * torch.randn(ch, 2880000, device="cuda")).round()
out = torch.empty_like(x)
native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(x), native.ptr(out), 3, 0.6, 0, native.stream_ptr()), "enhance")
torch.cuda.synchronize()
L.egr_fatllama_trace_once.restype = C.c_int
L.egr_fatllama_trace_once.argtypes = [C.c_void_p, C.c_void_p]
print("channels per launch:", ch, flush=True)
native.check(L.egr_fatllama_trace_once(C.c_void_p(plan), native.stream_ptr()), "trace")
