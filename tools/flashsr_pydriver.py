"""Operator-by-operator Python walk of the FlashSR graph: TEST / ABLATION tooling, not part of the product.

`PyDriverEngine` subclasses the product's FlashSREngine (which only owns the C-ABI handle) and drives the SAME graph one library
operator at a time from Python (hundreds of ctypes calls per pass, torch owning the buffers): per-stage taps, single operators
with the engine's packed weights, the strict f32-MFMA / no-Winograd modes.  It shares the packing kernels of
csrc/egr_flashsr_pack.hip with the handle, so both hold identical operands, and tests/test_gpu_flashsr_capi.py holds its output
bit-for-bit equal to egr_flashsr_forward.  Nothing under comfyui-egregora-audio-super-resolution_amd/ imports this file.
"""
import ctypes as C
import math
import os
import sys
from pathlib import Path
from typing import Dict, List, Optional

import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
from packload import load_pack  # noqa: E402

load_pack()
from egregora_amd import device_ops, flashsr_arch as arch, flashsr_engine as E, native  # noqa: E402
from egregora_amd.flashsr_engine import (ACT_LEAKY, ACT_LOGCLAMP, ACT_NONE, ACT_SILU, ACT_TANH, EW_ADD, EW_ADD_SCALE, EW_AXPBY,  # noqa: E402,F401
                                         EW_COPY, EW_SCALE, EW_SILU, _p)


class PyDriverEngine(E.FlashSREngine):
    _KEEP_PARAMS = True
    def __init__(self, cfg: arch.FlashSRConfig, params: Dict[str, torch.Tensor], device="cuda"):
        super().__init__(cfg, params, device)
        self.flops = 0.0            # dense-contraction flops of the last forward (per call, all rows)
        self.count_flops = False
        self.prof = None            # when a list: (kind, flops, start_event, end_event) per MFMA kernel launch
        self.blocks = arch.unet_blocks(cfg)
        # packed fp32 weights, bf16x3 splits, Winograd U, folded time embedding: built on first use of self.w / w3 / wz / wshape
        self._w: Dict[str, torch.Tensor] = {}
        self._wz: Dict[str, int] = {}              # floats per component of the z-stacked Winograd packs
        self._w3: Dict[str, torch.Tensor] = {}     # three-way bf16 splits of self.w entries (egr_split3_pack)
        self._wshape: Dict[str, tuple] = {}
        self._packed = False
        self._g_dev = {}
        nb = cfg.n_fft // 2 + 1
        self.ldm = ((nb + 15) // 16) * 16
        self.alpha, self.sigma = arch.cosine_alpha_sigma(cfg, cfg.t_steps - 1)
        self._gn_ws = {}            # GroupNorm scratch per stream

    def _ensure_packed(self):
        if self._packed:
            return
        self._packed = True         # first: the packing code below goes through the same properties
        cfg = self.cfg
        nb = cfg.n_fft // 2 + 1
        self._pack(self._params)
        self._w["mel_fb"] = self._pack_dev(self.mel_fb, 0, nb, cfg.n_mels, nb, cfg.n_mels, 1, 1)
        self._split3("mel_fb")
        self._fold_time_embedding()

    @property
    def w(self):
        self._ensure_packed()
        return self._w

    @property
    def w3(self):
        self._ensure_packed()
        return self._w3

    @property
    def wz(self):
        self._ensure_packed()
        return self._wz

    @property
    def wshape(self):
        self._ensure_packed()
        return self._wshape

    def _split3(self, key: str):
        """self.w3[key] = [slabs][3][Cout][16] bf16 split of the packed fp32 weight self.w[key] ([..., slabs, Cout, 16])."""
        if self.mfma != "bf16x3":
            return
        wp = self.w[key]
        Co = wp.shape[-2]
        ns = wp.numel() // (Co * 16)
        w3 = torch.empty(ns * 3 * Co * 16, dtype=torch.bfloat16, device=self.dev)
        native.check(self.L.egr_split3_pack(_p(wp), _p(w3), ns, Co, self._st()), "egr_split3_pack")
        self.w3[key] = w3

    def _s3(self, key, Cin, x):
        """The split weights for this call, or None when the bf16x3 kernel does not apply (Cin % 16, alignment, mode)."""
        if self.mfma != "bf16x3" or key is None or Cin % 16 != 0 or x.data_ptr() % 16 != 0:
            return None
        return self.w3.get(key)

    # ------------------------------------------------------------------ weight packing (kernels shared with the C-ABI handle)
    @staticmethod
    def pack_matrix(w2: torch.Tensor) -> torch.Tensor:
        """[K][Cout] -> the kernels' slab-major layout [ceil(K/16)][Cout][16] in plain torch (tests build operands with it)."""
        K, Co = w2.shape
        Kp = ((K + 15) // 16) * 16
        if Kp != K:
            w2 = torch.cat([w2, w2.new_zeros(Kp - K, Co)], 0)
        return w2.reshape(Kp // 16, 16, Co).permute(0, 2, 1).contiguous()

    def _pack_dev(self, src: torch.Tensor, layout: int, K: int, N: int, Ci: int, Co: int, KH: int, KW: int) -> torch.Tensor:
        """egr_pack_weight: torch layout -> the kernels' slab-major [ceil(K/16)][N][16] (k contiguous, zero padded)."""
        dst = torch.empty(((K + 15) // 16, N, 16), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_pack_weight(_p(src), _p(dst), layout, K, N, Ci, Co, KH, KW, self._st()), "egr_pack_weight")
        return dst

    def add_upsample_phases(self, key: str, v: torch.Tensor):
        """nearest-2x upsample followed by a 3x3 conv == four 2x2 convs on the low-res input, one per output phase
        (a, b): taps that read the same source pixel are pre-summed (egr_phase_weights).  Registers key + '.ph{a}{b}'."""
        v = v.detach().float().contiguous().to(self.dev)                 # [Co,Ci,3,3]
        Co, Ci = v.shape[:2]
        ph = torch.empty((4, Co, Ci, 2, 2), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_phase_weights(_p(v), _p(ph), Co, Ci, self._st()), "egr_phase_weights")
        for a in (0, 1):
            for b in (0, 1):
                self.add_weight(f"{key}.ph{a}{b}", ph[2 * a + b])

    # F(2x2,3x3) paid from 256 channels (its transforms move 4x the tensor); F(4x4,3x3) moves 2.25x and pays from 128

    _G2 = [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]]

    def _g4(self):
        """G of the library's F(4x4,3x3) scheme (points 0, +-3/4, +-3/2, inf; csrc/egr_nn_wino4.hip)."""
        buf = (C.c_double * 18)()
        native.check(self.L.egr_winograd4_g(buf), "egr_winograd4_g")
        return [[buf[3 * j + k] for k in range(3)] for j in range(6)]

    def add_winograd(self, key: str, v: torch.Tensor):
        """U = G g G^T in float64 for Winograd F(2x2,3x3) (16 [Cin][Cout] matrices, key + '.wino') and, when enabled,
        F(4x4,3x3) (36 matrices, key + '.wino4'), each matrix packed slab-major like a 1x1 conv weight (egr_winograd_pack_u)."""
        v = v.detach().float().contiguous().to(self.dev)                # [Co,Ci,3,3]
        Co, Ci = v.shape[:2]
        for suffix, Gm in ((".wino", self._G2),) + (((".wino4", self._g4()),) if self.WINO_F4 else ()):
            n = len(Gm)
            if n not in self._g_dev:
                self._g_dev[n] = torch.tensor(Gm, dtype=torch.float64, device=self.dev).contiguous()
            zf = ((Ci + 15) // 16) * Co * 16
            packed = torch.empty((n * n, zf), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_winograd_pack_u(_p(v), _p(packed), _p(self._g_dev[n]), n, Co, Ci, self._st()), "egr_winograd_pack_u")
            self.w[key + suffix] = packed.view(n * n, (Ci + 15) // 16, Co, 16)   # [n*n][Kp/16][Co][16]
            self.wz[key + suffix] = zf                                  # floats per component
            if Ci % 16 == 0:
                self._split3(key + suffix)
                if (key + suffix) in self.w3:                           # the fp32 pack is not needed once split
                    self.w[key + suffix] = self.w[key + suffix][:0]

    def _conv_winograd(self, x, key, act, res, bias_t, gn=None):
        B, H, W, Cin = x.shape
        Cout = self.wshape[key + ".weight"][3]
        f4 = (key + ".weight.wino4") in self.w and H % 4 == 0 and W % 4 == 0
        wkey = key + (".weight.wino4" if f4 else ".weight.wino")
        nz, ts = (36, 4) if f4 else (16, 2)
        fn_in, fn_out = (self.L.egr_winograd4_input, self.L.egr_winograd4_output) if f4 else \
            (self.L.egr_winograd_input, self.L.egr_winograd_output)
        TH, TW = (H + ts - 1) // ts, (W + ts - 1) // ts
        P = B * TH * TW
        V = torch.empty((nz, P, Cin), dtype=torch.float32, device=self.dev)
        gsc, gsh, gsilu = gn if gn is not None else (None, None, 0)
        native.check(fn_in(_p(x), _p(gsc), _p(gsh), gsilu, B, H, W, Cin, _p(V), self._st()), "egr_winograd_input")
        Mx = torch.empty((nz, P, Cout), dtype=torch.float32, device=self.dev)
        zw = self.wz[wkey]
        fl = nz * 2.0 * P * Cin * Cout
        ev = self._prof_begin()
        w3 = self._s3(wkey, Cin, V)
        if w3 is not None:
            native.check(self.L.egr_conv_s3(_p(V), _p(w3), _p(None), _p(None), _p(None), _p(Mx), P, 1, 1, Cin, 1, 1, Cout, 1, 1, 1,
                                            1, 0, 0, 0, 0, 0.0, 1, 1, 0, 0, 1, 1, nz, P * Cin, zw * 3 // 8, P * Cout,
                                            self._st()), "egr_conv_s3(winograd)")
        else:
            native.check(self.L.egr_gemm_zbatched(_p(V), _p(self.w[wkey]), _p(Mx), nz, P, Cin, Cout, P * Cin, zw, P * Cout,
                                                  self._st()), "egr_gemm_zbatched")
        if ev is not None:
            kind = self._kind(P, Cin, Cout, w3 is not None)
            if w3 is not None and Cout > 64:        # s3_zs_nzb (csrc/egr_nn_gemm_s3.hip): z-streamed when >= 2 z per workgroup
                bn = 256 if (Cout >= 256 and Cout % 256 == 0) else 128
                tiles = ((P + 127) // 128) * ((Cout + bn - 1) // bn)
                groups = min(max((2048 + tiles - 1) // tiles, 1), nz)
                if (nz + groups - 1) // groups >= 2 and not (bn == 256 and Cin > 256):
                    kind = f"k_conv_s3<128, {bn}, 1, true>"
            self._prof_end(ev, kind, fl, (B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, nz))
        if self.count_flops:
            self.flops += fl
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.dev)
        bt = bias_t if bias_t is not None else self.w.get(key + ".bias")
        G = self.cfg.gn_groups
        if f4 and self.GN_PARTIALS and Cout % G == 0 and (Cout // G) % 4 == 0:
            # the output transform also leaves per-thread (sum, sum of squares): a GroupNorm of y then reads 1/32 of its bytes
            part = torch.empty((P, Cout // 4, 2), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_winograd4_output_stats(_p(Mx), _p(bt), _p(res), _p(y), B, H, W, Cout,
                                                           1 if act == ACT_SILU else 0, _p(part), self._st()),
                         "egr_winograd4_output_stats")
            y._egr_gn_partials = (part, TH * TW)
        else:
            native.check(fn_out(_p(Mx), _p(bt), _p(res), _p(y), B, H, W, Cout, 1 if act == ACT_SILU else 0, self._st()),
                         "egr_winograd_output")
        return y


    def add_weight(self, key: str, v: torch.Tensor):
        """Register a weight given in torch layout; self.w[key] holds the packed tensor, self.wshape[key] the
        logical (KH, KW, Cin, Cout)."""
        v = v.detach().float().contiguous().to(self.dev)                # packed on the device
        if v.dim() == 4:                                                # conv2d [Co,Ci,kh,kw]
            Co, Ci, kh, kw = v.shape
            pk, shp = self._pack_dev(v, 0, kh * kw * Ci, Co, Ci, Co, kh, kw), (kh, kw, Ci, Co)
        elif key.startswith("voc.ups."):                                # convT1d [Ci,Co,k] -> GEMM [Ci][k*Co]
            Ci, Co, k = v.shape
            pk, shp = self._pack_dev(v, 1, Ci, k * Co, Ci, Co, 1, k), (1, 1, Ci, k * Co)
        elif v.dim() == 3:                                              # conv1d [Co,Ci,k]
            Co, Ci, k = v.shape
            pk, shp = self._pack_dev(v, 0, k * Ci, Co, Ci, Co, 1, k), (1, k, Ci, Co)
        else:                                                           # linear [Co,Ci]
            Co, Ci = v.shape
            pk, shp = self._pack_dev(v, 0, Ci, Co, Ci, Co, 1, 1), (1, 1, Ci, Co)
        self.w[key] = pk
        self.wshape[key] = shp
        self.w3.pop(key, None)
        if shp[2] % 16 == 0:
            self._split3(key)

    def _pack(self, P):
        self._wshape.clear()
        for k, v in P.items():
            if k.endswith(".weight") and v.dim() >= 2:
                self.add_weight(k, v)
                if ".upsample.conv." in k or (k.startswith("unet.") and ".up.conv." in k):
                    self.add_upsample_phases(k, v)
                if self.thin and v.dim() == 4 and v.shape[2] * v.shape[3] * v.shape[0] <= 32 and v.shape[1] % 16 == 0:
                    Co, Ci, kh, kw = v.shape                             # few outputs: 1x1 contraction onto per-tap products
                    self.add_weight(k + ".taps", v.permute(2, 3, 0, 1).reshape(kh * kw * Co, Ci))
                if v.dim() == 4 and v.shape[2] == 3 and v.shape[3] == 3 and min(v.shape[0], v.shape[1]) >= self.WINO_MIN_CH \
                        and "downsample" not in k and ".down.conv" not in k and "upsample" not in k and ".up.conv" not in k:
                    self.add_winograd(k, v)
            else:
                self.w[k] = v.detach().float().contiguous().to(self.dev)

    # ------------------------------------------------------------------ op wrappers

    def conv(self, x, wkey, B, H, W, Cin, OH, OW, Cout, KH, KW, stride=1, dil=1, pad_t=0, pad_l=0, up2=0, act=ACT_NONE,
             bias=True, bias_t=None, res=None, act_param=0.0, w=None, w3key=None):
        y = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=self.dev)
        wt = w if w is not None else self.w[wkey + ".weight"]
        bt = bias_t if bias_t is not None else (self.w.get(wkey + ".bias") if bias else None)
        fl = 2.0 * B * OH * OW * Cout * KH * KW * Cin
        ev = self._prof_begin()
        w3 = self._s3((wkey + ".weight") if w is None else w3key, Cin, x)
        if w3 is not None:
            native.check(self.L.egr_conv_s3(_p(x), _p(w3), _p(bt), _p(None), _p(res), _p(y), B, H, W, Cin, OH, OW, Cout, KH, KW,
                                            stride, dil, pad_t, pad_l, up2, act, float(act_param), 1, 1, 0, 0, OH, OW, 1, 0, 0, 0,
                                            self._st()), "egr_conv_s3")
        else:
            native.check(self.L.egr_conv_nhwc(_p(x), _p(wt), _p(bt), _p(None), _p(res), _p(y), B, H, W, Cin, OH, OW, Cout, KH,
                                              KW, stride, dil, pad_t, pad_l, up2, act, float(act_param), self._st()),
                         "egr_conv_nhwc")
        if ev is not None:
            vec = Cin % 16 == 0 and x.data_ptr() % 16 == 0
            kind = self._kind(B * OH * OW, Cin, Cout, w3 is not None, vec, KH * KW * Cin)
            if (w3 is not None and H == 1 and KH == 1 and KW >= 2 and stride == 1 and not up2 and OW == W and W % 128 == 0
                    and dil * (KW - 1) <= 50 and 2 * pad_l == dil * (KW - 1)):       # launch_conv1d_s3's conditions
                kind = f"k_conv1d_s3<{128 if Cout > 64 else (64 if Cout > 32 else 32)}, {32 if Cin % 32 == 0 else 16}>"
            self._prof_end(ev, kind, fl, (B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, up2))
        if self.count_flops:
            self.flops += fl
        return y

    @staticmethod
    def _kind(M, Cin, Cout, s3, vec=True, K=None):
        """Name of the kernel instantiation a contraction lands on (same selection as conv_launch, csrc/egr_nn_gemm.hip)."""
        bn = 128 if Cout > 64 else (64 if Cout > 32 else 32)
        if s3:
            if Cout >= 256 and Cout % 256 == 0:         # s3_bn
                bn = 256
            if K is not None and (K + 15) // 16 < 32:   # short-K, under-filled grid: narrower column tiles
                while bn > 64 and ((M + 127) // 128) * ((Cout + bn - 1) // bn) < 256:
                    bn >>= 1
            bm = 256 if (bn == 128 and ((M + 255) // 256) * ((Cout + 127) // 128) >= 1024) else 128
            return f"k_conv_s3<{bm}, {bn}, 1, false>"        # <BM, BN, PF, ZS> as rocprofv3 prints the instantiation
        return f"k_conv_igemm<{bn}, {'true' if vec else 'false'}>"

    def _prof_begin(self):
        if self.prof is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                      # torch's current stream == the stream the kernel is launched on
        return ev

    def _prof_end(self, ev, kind, flops, shape=None):
        if ev is None:
            return
        e2 = torch.cuda.Event(enable_timing=True)
        e2.record()
        self.prof.append((kind, flops, ev, e2, shape))

    def prof_summary(self):
        """{kind: (launches, total_flops, total_ms)} from the events collected while self.prof was a list."""
        torch.cuda.synchronize()
        out = {}
        for kind, fl, a, b, _ in self.prof or []:
            n, f, t = out.get(kind, (0, 0.0, 0.0))
            out[kind] = (n + 1, f + fl, t + a.elapsed_time(b))
        return out

    def conv3(self, x, key, stride=1, up2=0, act=ACT_NONE, res=None, pad=1, bias_t=None):
        B, H, W, Cin = x.shape
        Cout = self.wshape[key + ".weight"][3]
        if up2 and (key + ".weight.ph00") in self.w and stride == 1 and pad == 1 and res is None and bias_t is None:
            return self._conv_up2_phases(x, key, act)
        if (key + ".weight.wino") in self.w and not up2 and stride == 1 and pad == 1 and act in (ACT_NONE, ACT_SILU) \
                and H % 2 == 0 and W % 2 == 0 and Cin % 16 == 0:
            return self._conv_winograd(x, key, act, res, bias_t)
        plain = not up2 and stride == 1 and pad == 1 and act == ACT_NONE and res is None and bias_t is None
        if plain and (key + ".weight.taps") in self.w and self._s3(key + ".weight.taps", Cin, x) is not None:
            P = self.conv(x, None, B, H, W, Cin, H, W, 9 * Cout, 1, 1, bias=False, w=self.w[key + ".weight.taps"],
                          w3key=key + ".weight.taps")
            y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_tap_gather(_p(P), _p(self.w.get(key + ".bias")), _p(y), B, H, W, 3, 3, Cout, 1, 1, self._st()),
                         "egr_tap_gather")
            return y
        if plain and self.thin and Cin == 1 and Cout % 4 == 0 and 256 % (Cout // 4) == 0:
            y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_conv_cin1(_p(x), _p(self.w[key + ".weight"]), _p(self.w.get(key + ".bias")), _p(y), B, H, W, Cout,
                                              3, 3, 1, 1, self._st()), "egr_conv_cin1")
            if self.count_flops:
                self.flops += 2.0 * B * H * W * Cout * 9
            return y
        LH, LW = (2 * H, 2 * W) if up2 else (H, W)
        OH, OW = (LH // stride, LW // stride)
        return self.conv(x, key, B, H, W, Cin, OH, OW, Cout, 3, 3, stride, 1, pad, pad, up2, act, res=res, bias_t=bias_t)

    def _conv_up2_phases(self, x, key, act):
        B, H, W, Cin = x.shape
        Cout = self.wshape[key + ".weight"][3]
        y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float32, device=self.dev)
        bt = self.w.get(key + ".bias")
        for a in (0, 1):
            for b in (0, 1):
                fl = 2.0 * B * H * W * Cout * 4 * Cin
                ev = self._prof_begin()
                w3 = self._s3(f"{key}.weight.ph{a}{b}", Cin, x)
                if w3 is not None:
                    native.check(self.L.egr_conv_s3(_p(x), _p(w3), _p(bt), _p(None), _p(None), _p(y), B, H, W, Cin, H, W, Cout, 2, 2,
                                                    1, 1, 1 - a, 1 - b, 0, act, 0.0, 2, 2, a, b, 2 * H, 2 * W, 1, 0, 0, 0,
                                                    self._st()), "egr_conv_s3(placed)")
                else:
                    native.check(self.L.egr_conv_nhwc_placed(_p(x), _p(self.w[f"{key}.weight.ph{a}{b}"]), _p(bt), _p(None),
                                                             _p(None), _p(y), B, H, W, Cin, H, W, Cout, 2, 2, 1, 1, 1 - a, 1 - b,
                                                             0, act, 0.0, 2, 2, a, b, 2 * H, 2 * W, self._st()),
                                 "egr_conv_nhwc_placed")
                if ev is not None:
                    self._prof_end(ev, self._kind(B * H * W, Cin, Cout, w3 is not None, Cin % 16 == 0, 4 * Cin), fl,
                                   (B, H, W, Cin, H, W, Cout, 2, 2, 1, 1, 2))
                if self.count_flops:
                    self.flops += fl
        return y

    def conv1x1(self, x, key, res=None, act=ACT_NONE):
        B, H, W, Cin = x.shape
        Cout = self.wshape[key + ".weight"][3]
        return self.conv(x, key, B, H, W, Cin, H, W, Cout, 1, 1, res=res, act=act)

    def linear(self, x2, key, res=None, act=ACT_NONE, bias=True):
        rows, Cin = x2.shape
        Cout = self.wshape[key + ".weight"][3]
        y = self.conv(x2, key, rows, 1, 1, Cin, 1, 1, Cout, 1, 1, res=res, act=act, bias=bias)
        return y.view(rows, Cout)

    def conv1d(self, x, key, k, stride=1, dil=1, pad=0, act=ACT_NONE, res=None):
        B, L, Cin = x.shape
        Cout = self.wshape[key + ".weight"][3]
        OL = (L + 2 * pad - dil * (k - 1) - 1) // stride + 1
        y = self.conv(x, key, B, 1, L, Cin, 1, OL, Cout, 1, k, stride, dil, 0, pad, 0, act, res=res)
        return y.view(B, OL, Cout)

    def _gn_scratch(self, need):
        key = torch.cuda.current_stream().cuda_stream
        ws = self._gn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._gn_ws[key] = torch.empty(int(need) + 1024, dtype=torch.uint8, device=self.dev)
        return ws

    def groupnorm(self, x, key, eps, silu):
        B = x.shape[0]
        Cc = x.shape[-1]
        HW = x.numel() // (B * Cc)
        G = self.cfg.gn_groups
        need = self.L.egr_groupnorm_workspace_bytes(B, Cc, G)
        ws = self._gn_scratch(need)
        y = torch.empty_like(x)
        native.check(self.L.egr_groupnorm_nhwc(_p(x), _p(self.w[key + ".weight"]), _p(self.w[key + ".bias"]), _p(y), B, HW,
                                               Cc, G, eps, 1 if silu else 0, _p(ws), self._st()),
                     "egr_groupnorm_nhwc")
        return y

    def gn_coeff(self, x, key, eps):
        """GroupNorm statistics only -> (scale [B,C], shift [B,C]); the consumer conv applies them while loading."""
        B = x.shape[0]
        Cc = x.shape[-1]
        HW = x.numel() // (B * Cc)
        G = self.cfg.gn_groups
        need = self.L.egr_groupnorm_workspace_bytes(B, Cc, G)
        ws = self._gn_scratch(need)
        sc = torch.empty((B, Cc), dtype=torch.float32, device=self.dev)
        sh = torch.empty((B, Cc), dtype=torch.float32, device=self.dev)
        gp = getattr(x, "_egr_gn_partials", None)
        if gp is not None:           # x came out of egr_winograd4_output_stats: reduce its partials instead of re-reading x
            part, tiles = gp
            stats = torch.empty((B, G, 2), dtype=torch.float64, device=self.dev)
            native.check(self.L.egr_groupnorm_stats_from_partials(_p(part), B, tiles, Cc, G, _p(stats), self._st()),
                         "egr_groupnorm_stats_from_partials")
            native.check(self.L.egr_groupnorm_coeff_from_stats(_p(stats), _p(self.w[key + ".weight"]), _p(self.w[key + ".bias"]), B,
                                                               HW, Cc, G, eps, _p(sc), _p(sh), self._st()),
                         "egr_groupnorm_coeff_from_stats")
            return sc, sh
        native.check(self.L.egr_groupnorm_coeff(_p(x), _p(self.w[key + ".weight"]), _p(self.w[key + ".bias"]), B, HW, Cc, G,
                                                eps, _p(ws), _p(sc), _p(sh), self._st()), "egr_groupnorm_coeff")
        return sc, sh

    def gn_conv3(self, x, norm_key, eps, conv_key, res=None, bias_t=None):
        """conv3x3(silu(groupnorm(x))) with the normalisation fused into the conv's input path when possible."""
        B, H, W, Cin = x.shape
        Cout = self.wshape[conv_key + ".weight"][3]
        wino = (conv_key + ".weight.wino") in self.w and H % 2 == 0 and W % 2 == 0
        # Fusing into the Winograd input transform is free (HBM-bound kernel).  Fusing into the direct conv's loader
        # recomputes the SiLU once per tap and measured slower (conv 104 -> 91 TFLOP/s), so it is opt-in ("all").
        fused_ok = self.FUSE_GN != "0" and Cin % 16 == 0 and Cin % self.cfg.gn_groups == 0 and (wino or self.FUSE_GN == "all")
        if not fused_ok:
            return self.conv3(self.groupnorm(x, norm_key, eps, True), conv_key, res=res, bias_t=bias_t)
        sc, sh = self.gn_coeff(x, norm_key, eps)
        if wino:
            return self._conv_winograd(x, conv_key, ACT_NONE, res, bias_t, gn=(sc, sh, 1))
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=self.dev)
        bt = bias_t if bias_t is not None else self.w.get(conv_key + ".bias")
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        ev = self._prof_begin()
        native.check(self.L.egr_conv_nhwc_gn(_p(x), _p(sc), _p(sh), 1, _p(self.w[conv_key + ".weight"]), _p(bt), _p(res), _p(y),
                                             B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 1, ACT_NONE, self._st()), "egr_conv_nhwc_gn")
        if ev is not None:
            bn = 128 if Cout > 64 else (64 if Cout > 32 else 32)
            self._prof_end(ev, f"k_conv_igemm<{bn}, true>", fl, (B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, 0))
        if self.count_flops:
            self.flops += fl
        return y


    def layernorm(self, x2, key):
        rows, Cc = x2.shape
        y = torch.empty_like(x2)
        native.check(self.L.egr_layernorm_rows(_p(x2), _p(self.w[key + ".weight"]), _p(self.w[key + ".bias"]), _p(y), rows,
                                               Cc, 1e-5, self._st()), "egr_layernorm_rows")
        return y

    def eltwise(self, a, b, op, s0=0.0, s1=0.0):
        y = torch.empty_like(a)
        native.check(self.L.egr_eltwise(_p(a), _p(b), _p(y), a.numel(), op, s0, s1, self._st()), "egr_eltwise")
        return y

    def attention(self, q, k, v, B, T, Cc, heads):
        """q,k,v [B*T, C] -> [B*T, C]; softmax(q k^T / sqrt(d)) v per head.  bf16x3 mode: both GEMMs as A * B^T on the bf16 pipe
        (V is transposed per row first, [B][T][C] -> [B][C][T], so that its head slices are K-contiguous)."""
        d = Cc // heads
        S = torch.empty((B, heads, T, T), dtype=torch.float32, device=self.dev)
        s3 = self.mfma == "bf16x3" and d % 16 == 0 and T % 16 == 0 and Cc % 4 == 0
        if s3:
            native.check(self.L.egr_bgemm_nt_s3(_p(q), _p(k), _p(S), B, heads, T, T, d, Cc, Cc, T, T * Cc, d, T * Cc, d,
                                                heads * T * T, T * T, d ** -0.5, self._st()), "egr_bgemm_nt_s3(QK^T)")
        else:
            native.check(self.L.egr_bgemm(_p(q), _p(k), _p(S), B, heads, T, T, d, Cc, Cc, T, T * Cc, d, T * Cc, d,
                                          heads * T * T, T * T, 1, d ** -0.5, self._st()), "egr_bgemm(QK^T)")
        native.check(self.L.egr_softmax_rows(_p(S), B * heads * T, T, self._st()), "egr_softmax_rows")
        o = torch.empty((B * T, Cc), dtype=torch.float32, device=self.dev)
        if s3:
            vt = torch.empty((B, Cc, T), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_transpose_batched(_p(v), _p(vt), B, T, Cc, self._st()), "egr_transpose_batched")
            native.check(self.L.egr_bgemm_nt_s3(_p(S), _p(vt), _p(o), B, heads, T, d, T, T, T, Cc, heads * T * T, T * T, Cc * T,
                                                d * T, T * Cc, d, 1.0, self._st()), "egr_bgemm_nt_s3(PV)")
        else:
            native.check(self.L.egr_bgemm(_p(S), _p(v), _p(o), B, heads, T, d, T, T, Cc, Cc, heads * T * T, T * T, T * Cc, d,
                                          T * Cc, d, 0, 1.0, self._st()), "egr_bgemm(PV)")
        if self.count_flops:
            self.flops += 4.0 * B * heads * T * T * d
        return o

    def snake(self, x, akey, bkey):
        B, L, Cc = x.shape
        y = torch.empty_like(x)
        native.check(self.L.egr_snake_aa(_p(x), _p(self.w[akey]), _p(self.w[bkey]), _p(self.filt), _p(y), B, L, Cc,
                                         self.cfg.aa_taps, self._st()), "egr_snake_aa")
        return y

    # ------------------------------------------------------------------ constant sub-graph: time embedding at t = T-1
    def _fold_time_embedding(self):
        emb = self.time_embedding_input()
        t = self.linear(emb, "unet.time_embed.0", act=ACT_SILU)
        t = self.linear(t, "unet.time_embed.2", act=ACT_SILU)            # silu(temb), shared by all res-blocks
        for name, cin, cout, attn in self.blocks:
            if name.endswith(".block"):
                base = f"unet.{name}"
                e = self.linear(t, base + ".res.emb")                    # [1, cout]
                # conv bias + time bias, both per output channel and identical for every row at fixed t
                self.w[base + ".res.in_conv.bias_t"] = self.eltwise(self.w[base + ".res.in_conv.bias"].view(1, -1), e,
                                                                    EW_ADD).view(-1)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ stages
    def log_mel(self, x):
        cfg = self.cfg
        B, L = x.shape
        rpad = (cfg.n_fft - cfg.hop) // 2
        t_valid = min(cfg.n_frames, (L + 2 * rpad - cfg.n_fft) // cfg.hop + 1)
        mag = torch.empty((B, cfg.n_frames, self.ldm), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_stft_frames(_p(x), B, L, cfg.n_fft, cfg.hop, rpad, cfg.n_frames, t_valid, self.ldm,
                                            _p(self.window), _p(mag), self._st()), "egr_stft_frames")
        mel = self.conv(mag, None, B * cfg.n_frames, 1, 1, self.ldm, 1, 1, cfg.n_mels, 1, 1, act=ACT_LOGCLAMP,
                        act_param=cfg.log_floor, bias=False, w=self.w["mel_fb"], w3key="mel_fb")
        return mel.view(B, cfg.n_frames, cfg.n_mels, 1)

    # input low-pass (lowpass_input=True): cutoff from the STFT energy, zero-phase 8th-order Chebyshev-I gain applied
    # in the frequency domain on the Fat-Llama transform passes (UPSTREAM-RECALL of FlashSR's cheby/filtfilt
    # pre-filter; edge handling differs from a time-domain filtfilt by construction).
    LP_PCT, LP_ORDER, LP_RIPPLE_DB = 0.985, 8, 0.05

    def lowpass(self, x):
        from egregora_amd import fatllama_engine as fe
        cfg = self.cfg
        B, L = x.shape
        rpad = (cfg.n_fft - cfg.hop) // 2
        T = (L + 2 * rpad - cfg.n_fft) // cfg.hop + 1
        nb = cfg.n_fft // 2 + 1
        mag = torch.empty((B, T, self.ldm), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_stft_frames(_p(x), B, L, cfg.n_fft, cfg.hop, rpad, T, T, self.ldm, _p(self.window), _p(mag),
                                            self._st()), "egr_stft_frames")
        cut = torch.empty((B,), dtype=torch.int32, device=self.dev)
        nbins = L // 2 + 1
        gain = torch.empty((B, nbins), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_lowpass_gain(_p(mag), B, T, self.ldm, nb, self.LP_PCT, float(cfg.sr), self.LP_ORDER,
                                             self.LP_RIPPLE_DB, nbins, _p(cut), _p(gain), self._st()), "egr_lowpass_gain")
        plan = fe._plan(L, B, 1, self.dev.index or 0)
        y = torch.empty_like(x)
        native.check(self.L.egr_spectral_gain(C.c_void_p(plan), _p(x), _p(gain), _p(y), self._st()), "egr_spectral_gain")
        self.last_cutoff_bins = cut
        return y

    def _vae_res(self, x, name):
        h = self.gn_conv3(x, name + ".norm1", 1e-6, name + ".conv1")
        sc = self.conv1x1(x, name + ".nin_shortcut") if (name + ".nin_shortcut.weight") in self.w else x
        return self.gn_conv3(h, name + ".norm2", 1e-6, name + ".conv2", res=sc)

    def _vae_attn(self, x, name):
        B, H, W, Cc = x.shape
        h = self.groupnorm(x, name + ".norm", 1e-6, False)
        q = self.conv1x1(h, name + ".q").view(B * H * W, Cc)
        k = self.conv1x1(h, name + ".k").view(B * H * W, Cc)
        v = self.conv1x1(h, name + ".v").view(B * H * W, Cc)
        o = self.attention(q, k, v, B, H * W, Cc, 1).view(B, H, W, Cc)
        return self.conv1x1(o, name + ".proj_out", res=x)

    def vae_encode(self, mel):
        cfg = self.cfg
        h = self.conv3(mel, "vae.encoder.conv_in")
        n = len(cfg.vae_mult)
        for lv in range(n):
            for b in range(cfg.vae_res):
                h = self._vae_res(h, f"vae.encoder.down.{lv}.block.{b}")
            if lv != n - 1:
                h = self.conv3(h, f"vae.encoder.down.{lv}.downsample.conv", stride=2, pad=0)
        h = self._vae_res(h, "vae.encoder.mid.block_1")
        h = self._vae_attn(h, "vae.encoder.mid.attn_1")
        h = self._vae_res(h, "vae.encoder.mid.block_2")
        h = self.conv3(self.groupnorm(h, "vae.encoder.norm_out", 1e-6, True), "vae.encoder.conv_out")
        mom = self.conv1x1(h, "vae.quant_conv")
        return mom[..., :cfg.z_ch].contiguous()

    def vae_decode(self, z):
        cfg = self.cfg
        h = self.conv1x1(z, "vae.post_quant_conv")
        h = self.conv3(h, "vae.decoder.conv_in")
        h = self._vae_res(h, "vae.decoder.mid.block_1")
        h = self._vae_attn(h, "vae.decoder.mid.attn_1")
        h = self._vae_res(h, "vae.decoder.mid.block_2")
        for lv in reversed(range(len(cfg.vae_mult))):
            for b in range(cfg.vae_res + 1):
                h = self._vae_res(h, f"vae.decoder.up.{lv}.block.{b}")
            if lv != 0:
                h = self.conv3(h, f"vae.decoder.up.{lv}.upsample.conv", up2=1)
        return self.conv3(self.groupnorm(h, "vae.decoder.norm_out", 1e-6, True), "vae.decoder.conv_out")

    def _unet_block(self, x, base, has_attn):
        cfg = self.cfg
        h = self.gn_conv3(x, base + ".res.in_norm", 1e-5, base + ".res.in_conv", bias_t=self.w[base + ".res.in_conv.bias_t"])
        sc = self.conv1x1(x, base + ".res.skip") if (base + ".res.skip.weight") in self.w else x
        x = self.gn_conv3(h, base + ".res.out_norm", 1e-5, base + ".res.out_conv", res=sc)
        if has_attn:
            B, H, W, Cc = x.shape
            T = H * W
            heads = Cc // cfg.head_dim
            t = self.conv1x1(self.groupnorm(x, base + ".st.norm", 1e-6, False), base + ".st.proj_in").view(B * T, Cc)
            for a in ("attn1", "attn2"):
                n_ = self.layernorm(t, f"{base}.st.{a}_ln")
                q = self.linear(n_, f"{base}.st.{a}.to_q", bias=False)
                k = self.linear(n_, f"{base}.st.{a}.to_k", bias=False)
                v = self.linear(n_, f"{base}.st.{a}.to_v", bias=False)
                o = self.attention(q, k, v, B, T, Cc, heads)
                t = self.linear(o, f"{base}.st.{a}.to_out", res=t)
            u = self.linear(self.layernorm(t, base + ".st.ff_ln"), base + ".st.ff.geglu")
            g = torch.empty((B * T, 4 * Cc), dtype=torch.float32, device=self.dev)
            native.check(self.L.egr_geglu(_p(u), _p(g), B * T, 4 * Cc, self._st()), "egr_geglu")
            t = self.linear(g, base + ".st.ff.out", res=t)
            x = self.conv1x1(t.view(B, H, W, Cc), base + ".st.proj_out", res=x)
        return x

    def concat(self, a, b):
        B, H, W, C1 = a.shape
        C2 = b.shape[3]
        y = torch.empty((B, H, W, C1 + C2), dtype=torch.float32, device=self.dev)
        native.check(self.L.egr_concat_channels(_p(a), _p(b), _p(y), B * H * W, C1, C2, self._st()), "egr_concat_channels")
        return y

    def unet(self, x):
        skips: List[torch.Tensor] = []
        h = x
        for name, cin, cout, attn in self.blocks:
            part, _, kind = name.split(".")
            base = f"unet.{name}"
            if kind == "conv_in":
                h = self.conv3(h, base)
                skips.append(h)
            elif kind == "down":
                h = self.conv3(h, base + ".conv", stride=2, pad=1)
                skips.append(h)
            elif kind == "up":
                h = self.conv3(h, base + ".conv", up2=1)
            else:
                if part == "out":
                    h = self.concat(h, skips.pop())
                h = self._unet_block(h, base, attn)
                if part == "in":
                    skips.append(h)
        return self.conv3(self.groupnorm(h, "unet.out_norm", 1e-5, True), "unet.out_conv")

    def _amp(self, h, j):
        cfg = self.cfg
        acc = None
        for ki, k in enumerate(cfg.voc_kernels):
            x = h
            for di, d in enumerate(cfg.voc_dils):
                b = f"voc.amp.{j}.{ki}.{di}"
                xt = self.snake(x, b + ".alpha1", b + ".beta1")
                xt = self.conv1d(xt, b + ".conv1", k, dil=d, pad=d * (k - 1) // 2)
                xt = self.snake(xt, b + ".alpha2", b + ".beta2")
                x = self.conv1d(xt, b + ".conv2", k, pad=(k - 1) // 2, res=x)
            if acc is None:
                acc = x
            elif ki + 1 < len(cfg.voc_kernels):
                acc = self.eltwise(acc, x, EW_ADD)
            else:                                            # last branch: the mean's scale rides on the last add
                return self.eltwise(acc, x, EW_ADD_SCALE, 1.0 / len(cfg.voc_kernels))
        return self.eltwise(acc, None, EW_SCALE, 1.0 / len(cfg.voc_kernels))

    def vocoder(self, mel_hat, wave):
        cfg = self.cfg
        B, T, Fm, _ = mel_hat.shape
        n = len(cfg.voc_rates)
        feats = []
        e = wave.view(B, -1, 1)
        for i, r in enumerate(reversed(cfg.voc_rates)):
            e = self.conv1d(e, f"voc.wave_enc.{i}", 2 * r + 1, stride=r, pad=r, act=ACT_LEAKY)
            feats.append(e)
        h = self.conv1d(mel_hat.view(B, T, Fm), "voc.conv_pre", 7, pad=3, res=feats[n - 1])
        for j, r in enumerate(cfg.voc_rates):
            kt = arch.up_kernel(r)
            Bc, Lin, Ci = h.shape
            wt = self.w[f"voc.ups.{j}.weight"]
            Co = self.wshape[f"voc.ups.{j}.weight"][3] // kt
            Y = self.conv(h, None, Bc * Lin, 1, 1, Ci, 1, 1, kt * Co, 1, 1, bias=False, w=wt, w3key=f"voc.ups.{j}.weight")
            out = torch.empty((Bc, Lin * r, Co), dtype=torch.float32, device=self.dev)
            add = feats[n - 2 - j] if j <= n - 2 else None
            native.check(self.L.egr_col2im_convtr1d(_p(Y), _p(self.w[f"voc.ups.{j}.bias"]), _p(add), _p(out), Bc, Lin,
                                                    Lin * r, Co, kt, r, (kt - r) // 2, self._st()), "egr_col2im_convtr1d")
            h = self._amp(out, j)
        h = self.snake(h, "voc.post.alpha", "voc.post.beta")
        y = self.conv1d(h, "voc.conv_post", 7, pad=3, act=ACT_TANH)
        return y.view(B, -1)

    def forward_rows(self, x: torch.Tensor, noise: torch.Tensor, stages: Optional[dict] = None,
                     lowpass: bool = False) -> torch.Tensor:
        """x [R, chunk] float32 CUDA, noise [R, h, w, z] (channels-last) -> y [R, chunk]."""
        x = x.contiguous()
        if lowpass:
            x = self.lowpass(x)
        mel = self.log_mel(x)
        z_c = self.vae_encode(mel)
        v = self.unet(self.concat(noise, z_c))
        z0 = self.eltwise(noise, v, EW_AXPBY, self.alpha, -self.sigma)
        mel_hat = self.vae_decode(z0)
        y = self.vocoder(mel_hat, x)
        if stages is not None:
            stages.update(mel=mel, z_cond=z_c, v=v, z0=z0, mel_hat=mel_hat, y=y)
        return y[:, :x.shape[1]]

    def flop_count_py(self, rows: int = 1) -> float:
        """Dense-contraction flops of one forward over `rows` rows (dry run of the Python walk with counting on)."""
        x = torch.zeros((rows, self.cfg.chunk), dtype=torch.float32, device=self.dev)
        self.count_flops, self.flops = True, 0.0
        self.forward_rows(x, self.noise(rows, None, 0))
        torch.cuda.synchronize()
        self.count_flops = False
        return self.flops
