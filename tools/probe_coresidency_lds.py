"""Bisect of the k_stft_frames x k_conv_s3 co-residency corruption: does it depend on the foreign kernel's LDS
footprint?  k_conv_s3<128,256> holds 73.7 KB of static LDS (> 64 KB), <128,128> 49 KB.  Run twice: default and EGR_S3_BN256=0."""
import os, sys; sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E, streams
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); e = PyDriverEngine(cfg, A.init_params(cfg, 0))
x = 0.2 * torch.randn(26, cfg.chunk, device='cuda')
side = streams.side_streams(3)
B3 = ((9, 17), (17, 26), (0, 9))
def stft_only(xs):
    B, L = xs.shape
    rpad = (cfg.n_fft - cfg.hop) // 2
    t_valid = min(cfg.n_frames, (L + 2 * rpad - cfg.n_fft) // cfg.hop + 1)
    mag = torch.empty((B, cfg.n_frames, e.ldm), dtype=torch.float32, device=e.dev)
    E.native.check(e.L.egr_stft_frames(E._p(xs.contiguous()), B, L, cfg.n_fft, cfg.hop, rpad, cfg.n_frames, t_valid, e.ldm, E._p(e.window), E._p(mag), e._st()), "stft")
    return mag
mags = {b: stft_only(x[b[0]:b[1]]) for b in B3}
mref = {b: mags[b].clone() for b in B3}
torch.cuda.synchronize()
def conv(b):
    B = b[1] - b[0]
    return e.conv(mags[b], None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
def run_mixed2():
    cur = torch.cuda.current_stream(); ready = cur.record_event()
    outs = {}
    for i, b in enumerate(B3):
        s = side[i % len(side)]; s.wait_event(ready)
        with torch.cuda.stream(s):
            if i == 0:
                for _ in range(6): conv(b)
            else:
                outs[b] = [stft_only(x[b[0]:b[1]]) for _ in range(6)]
    torch.cuda.synchronize()
    return outs
nbad = 0
for rep in range(40):
    o = run_mixed2()
    nbad += any(float((y - mref[b]).abs().max()) > 0 for b in o for y in o[b])
print("EGR_S3_BN256=%s n_side=%d : stft next to foreign s3 convs: bad runs of 40: %d" % (os.environ.get("EGR_S3_BN256", "1"), len(side), nbad))
