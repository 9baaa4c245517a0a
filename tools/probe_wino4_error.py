"""Dev probe: where the F(4x4,3x3) path's error comes from (V, GEMM, output transform), against float64."""
import sys; sys.path.insert(0, '.')
import ctypes as C, math, torch
from packload import load_pack; load_pack()
from egregora_amd import native, flashsr_arch as A, flashsr_engine as E
from flashsr_pydriver import PyDriverEngine
cfg = A.tiny_config(); e = PyDriverEngine(cfg, A.init_params(cfg, 0)); L = e.L
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
BT = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=torch.float64)
G = torch.tensor(e._g4(), dtype=torch.float64)
a_, b_ = 0.75, 1.5
BT = torch.tensor([[a_*a_*b_*b_,0,-(a_*a_+b_*b_),0,1,0],[0,-a_*b_*b_,-b_*b_,a_,1,0],[0,a_*b_*b_,-b_*b_,-a_,1,0],[0,-a_*a_*b_,-a_*a_,b_,1,0],[0,a_*a_*b_,-a_*a_,-b_,1,0],[0,a_*a_*b_*b_,0,-(a_*a_+b_*b_),0,1]], dtype=torch.float64)
AT = torch.tensor([[1,1,1,1,1,0],[0,a_,-a_,b_,-b_,0],[0,a_**2,a_**2,b_**2,b_**2,0],[0,a_**3,-a_**3,b_**3,-b_**3,1]], dtype=torch.float64)
AT = torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=torch.float64)
g = torch.Generator().manual_seed(45)
for (B, H, W, Ci, Co) in [(1, 16, 16, 256, 256), (1, 8, 8, 1024, 512)]:
    x = torch.randn(B, Ci, H, W, generator=g); w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    TH, TW = H // 4, W // 4; P = B * TH * TW
    V = torch.empty(36, P, Ci, device='cuda')
    native.check(L.egr_winograd4_input(p(xh), p(None), p(None), 0, B, H, W, Ci, p(V), e._st()), "in")
    tiles = torch.nn.functional.pad(x.double(), (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)        # B,C,th,tw,6,6
    V64 = torch.einsum('ik,bctukl,jl->ijbtuc', BT, tiles, BT).reshape(36, P, Ci)
    print("V err (max rel to max |V|):", float((V.cpu().double() - V64).abs().max() / V64.abs().max()), "max|V|", float(V64.abs().max()))
    U = torch.einsum('ik,ockl,jl->ijco', G, w.double(), G).reshape(36, Ci, Co).float()
    M64 = torch.einsum('zpc,zco->zpo', V.cpu().double(), U.double())
    e.add_weight("w4.weight", w); e.add_winograd("w4.weight", w)
    w3 = e.w3["w4.weight.wino4"]; zw = e.wz["w4.weight.wino4"]
    Mx = torch.empty(36, P, Co, device='cuda')
    native.check(L.egr_conv_s3(p(V), p(w3), p(None), p(None), p(None), p(Mx), P, 1, 1, Ci, 1, 1, Co, 1, 1, 1, 1, 0, 0, 0, 0, 0.0, 1, 1, 0, 0, 1, 1,
                               36, P * Ci, zw * 3 // 8, P * Co, e._st()), "gemm")
    print("M err given GPU V (rel to max|M|):", float((Mx.cpu().double() - M64).abs().max() / M64.abs().max()), "max|M|", float(M64.abs().max()))
    # per-component error
    pe = (Mx.cpu().double() - M64).abs().amax(dim=(1, 2)) / M64.abs().amax(dim=(1, 2))
    print("  per-component M rel err max:", float(pe.max()), "argmax", int(pe.argmax()))
    y = torch.empty(B, H, W, Co, device='cuda')
    native.check(L.egr_winograd4_output(p(Mx), p(None), p(None), p(y), B, H, W, Co, 0, e._st()), "out")
    Y64 = torch.einsum('ki,ijbtuo,lj->btuokl', AT, Mx.cpu().double().reshape(6, 6, B, TH, TW, Co), AT).permute(0, 3, 1, 4, 2, 5).reshape(B, Co, H, W)
    yy = y.permute(0, 3, 1, 2).cpu().double()
    print("out-transform err given GPU M:", float((yy - Y64).abs().max() / ref.abs().max()), " total err:", float((yy - ref).abs().max() / ref.abs().max()),
          " Y64(GPU M) vs ref:", float((Y64 - ref).abs().max() / ref.abs().max()))
