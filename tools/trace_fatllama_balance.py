"""dev: phase trace of the loop kernels on N = 2 x 500 x 2304 (251 row pairs: one workgroup per CU) vs N = 2 x 625 x 2304 (313)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import fatllama_engine as fe, native
L = native.lib()
L.egr_fatllama_trace_once.restype = C.c_int
L.egr_fatllama_trace_once.argtypes = [C.c_void_p, C.c_void_p]
for m1 in (500, 625, 512):
    n = 2 * m1 * 2304
    plan = fe._plan(n, 1, 1, 0, split=(m1, 2304, 1))
    x = (3000 * torch.randn(1, n, device="cuda")).round()
    out = torch.empty_like(x)
    native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(x), native.ptr(out), 3, 0.6, 0, native.stream_ptr()), "enhance")
    torch.cuda.synchronize()
    print("M1 =", m1, "row pairs =", m1 // 2 + 1, flush=True)
    native.check(L.egr_fatllama_trace_once(C.c_void_p(plan), native.stream_ptr()), "trace")
