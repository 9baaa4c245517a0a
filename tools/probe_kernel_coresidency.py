import sys, os; sys.path.insert(0, '.')
import torch
from packload import load_pack; load_pack()
from egregora_amd import flashsr_arch as A, flashsr_engine as E, streams
from flashsr_pydriver import PyDriverEngine
cfg = A.FlashSRConfig(); e = PyDriverEngine(cfg, A.init_params(cfg, 0))
x = 0.2 * torch.randn(26, cfg.chunk, device='cuda')
side = streams.side_streams(3)
print("n side", len(side))
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
B3 = ((9, 17), (17, 26), (0, 9))
ref = {b: e.log_mel(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
def run(fn):
    cur = torch.cuda.current_stream(); ready = cur.record_event()
    outs = {}
    for i, b in enumerate(B3):
        if mode == "default" and i == 2:
            outs[b] = fn(x[b[0]:b[1]])
        else:
            s = side[i % len(side)]; s.wait_event(ready)
            with torch.cuda.stream(s): outs[b] = fn(x[b[0]:b[1]])
    torch.cuda.synchronize()
    return outs
nbad = 0
for rep in range(60):
    o = run(e.log_mel)
    nbad += any(float((o[b] - ref[b]).abs().max()) > 0 for b in B3)
print(mode, "bad runs of 60:", nbad)
def stft_only(xs):
    B, L = xs.shape
    rpad = (cfg.n_fft - cfg.hop) // 2
    t_valid = min(cfg.n_frames, (L + 2 * rpad - cfg.n_fft) // cfg.hop + 1)
    mag = torch.empty((B, cfg.n_frames, e.ldm), dtype=torch.float32, device=e.dev)
    E.native.check(e.L.egr_stft_frames(E._p(xs.contiguous()), B, L, cfg.n_fft, cfg.hop, rpad, cfg.n_frames, t_valid, e.ldm, E._p(e.window), E._p(mag), e._st()), "stft")
    return mag
ref = {b: stft_only(x[b[0]:b[1]]).clone() for b in B3}
nbad = 0
for rep in range(60):
    o = run(stft_only)
    nbad += any(float((o[b] - ref[b]).abs().max()) > 0 for b in B3)
print(mode, "stft only: bad runs of 60:", nbad)
keep = []
def mel_keep(xs):
    B, L = xs.shape
    mag = stft_only(xs)
    keep.append(mag)
    mel = e.conv(mag, None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
    return mel.view(B, cfg.n_frames, cfg.n_mels, 1)
ref = {b: e.log_mel(x[b[0]:b[1]]).clone() for b in B3}
nbad = 0
for rep in range(40):
    keep.clear()
    o = run(mel_keep)
    nbad += any(float((o[b] - ref[b]).abs().max()) > 0 for b in B3)
print(mode, "mag kept alive: bad runs of 40:", nbad)
mags = {b: stft_only(x[b[0]:b[1]]) for b in B3}
torch.cuda.synchronize()
def conv_only_factory():
    def f(xs):
        B = xs.shape[0]
        b = next(bb for bb in B3 if bb[1] - bb[0] == B and xs.data_ptr() == x[bb[0]:bb[1]].data_ptr())
        mel = e.conv(mags[b], None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
        return mel.view(B, cfg.n_frames, cfg.n_mels, 1)
    return f
f = conv_only_factory()
ref = {b: f(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
nbad = 0; pat = []
for rep in range(40):
    o = run(f)
    bad = False
    for b in B3:
        d = (o[b] - ref[b]).abs().view(-1, cfg.n_mels)
        if float(d.max()) > 0:
            bad = True
            if len(pat) < 6:
                nzr = (d > 0).any(dim=1).nonzero().flatten().cpu().tolist()
                pat.append((b, len(nzr), nzr[:8], int((d > 0).sum())))
    nbad += bad
print(mode, "conv only (mag precomputed): bad runs of 40:", nbad, pat)
print("--- what do the wrong rows look like")
ref = {b: e.log_mel(x[b[0]:b[1]]).clone() for b in B3}
found = 0
for rep in range(40):
    o = run(e.log_mel)
    for b in B3:
        got = o[b].view(-1, cfg.n_mels); want = ref[b].view(-1, cfg.n_mels)
        bad_rows = ((got - want).abs() > 0).any(dim=1).nonzero().flatten().cpu().tolist()
        for r in bad_rows[:2]:
            g, w = got[r], want[r]
            print("rows", b, "row", r, "got[:6]", [round(float(v), 3) for v in g[:6]], "want[:6]", [round(float(v), 3) for v in w[:6]],
                  "log_floor", round(float(torch.log(torch.tensor(cfg.log_floor))), 3), "n cols differing", int(((g - w).abs() > 0).sum()),
                  "max rel diff", float(((g - w).abs() / (w.abs() + 1e-6)).max()))
            found += 1
    if found >= 6: break
print("--- writer / reader split")
def stft_then_torch_reader(xs):
    mag = stft_only(xs)
    return mag.clone()                     # torch copy kernel reads what stft wrote
ref = {b: stft_only(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
nbad = sum(any(float((o[b] - ref[b]).abs().max()) > 0 for b in B3) for o in (run(stft_then_torch_reader) for _ in range(40)))
print("stft -> torch reader: bad runs of 40:", nbad)
def torch_writer_then_conv(xs):
    B = xs.shape[0]
    b = next(bb for bb in B3 if bb[1] - bb[0] == B and xs.data_ptr() == x[bb[0]:bb[1]].data_ptr())
    mag = mags[b].clone()                  # torch copy kernel writes the conv's input
    mel = e.conv(mag, None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
    return mel.view(B, cfg.n_frames, cfg.n_mels, 1)
ref = {b: torch_writer_then_conv(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
nbad = sum(any(float((o[b] - ref[b]).abs().max()) > 0 for b in B3) for o in (run(torch_writer_then_conv) for _ in range(40)))
print("torch writer -> s3 conv: bad runs of 40:", nbad)
print("--- NaN prefill")
def nan_stft_conv(xs):
    B, L = xs.shape
    rpad = (cfg.n_fft - cfg.hop) // 2
    t_valid = min(cfg.n_frames, (L + 2 * rpad - cfg.n_fft) // cfg.hop + 1)
    mag = torch.empty((B, cfg.n_frames, e.ldm), dtype=torch.float32, device=e.dev)
    mag.fill_(float('nan'))
    E.native.check(e.L.egr_stft_frames(E._p(xs.contiguous()), B, L, cfg.n_fft, cfg.hop, rpad, cfg.n_frames, t_valid, e.ldm, E._p(e.window), E._p(mag), e._st()), "stft")
    mel = e.conv(mag, None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
    return mel.view(B, cfg.n_frames, cfg.n_mels, 1)
ref = {b: e.log_mel(x[b[0]:b[1]]).clone() for b in B3}
nbad = nnan = 0
for rep in range(40):
    o = run(nan_stft_conv)
    nbad += any(float((torch.nan_to_num(o[b]) - ref[b]).abs().max()) > 0 for b in B3)
    nnan += sum(int(torch.isnan(o[b]).sum()) for b in B3)
print("NaN prefill -> stft -> s3 conv: bad runs of 40:", nbad, "NaN outputs:", nnan)
print("--- conv on precomputed mag while OTHER streams run stft")
f = conv_only_factory()
ref = {b: f(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
def run_mixed():
    cur = torch.cuda.current_stream(); ready = cur.record_event()
    outs = {}
    for i, b in enumerate(B3):
        s = side[i % len(side)]; s.wait_event(ready)
        with torch.cuda.stream(s):
            if i == 0:
                outs[b] = [f(x[b[0]:b[1]]) for _ in range(6)]       # convs only, precomputed input
            else:
                for _ in range(6): stft_only(x[b[0]:b[1]])            # stft only
    torch.cuda.synchronize()
    return outs
nbad = 0
for rep in range(40):
    o = run_mixed()
    b = B3[0]
    nbad += any(float((y - ref[b]).abs().max()) > 0 for y in o[b])
print("conv (precomputed input) next to foreign stft: bad runs of 40:", nbad)
print("--- stft outputs while OTHER streams run s3 convs")
mref = {b: stft_only(x[b[0]:b[1]]).clone() for b in B3}
torch.cuda.synchronize()
def run_mixed2():
    cur = torch.cuda.current_stream(); ready = cur.record_event()
    outs = {}
    for i, b in enumerate(B3):
        s = side[i % len(side)]; s.wait_event(ready)
        with torch.cuda.stream(s):
            if i == 0:
                for _ in range(6): f(x[b[0]:b[1]])
            else:
                outs[b] = [stft_only(x[b[0]:b[1]]) for _ in range(6)]
    torch.cuda.synchronize()
    return outs
nbad = 0
for rep in range(40):
    o = run_mixed2()
    nbad += any(float((y - mref[b]).abs().max()) > 0 for b in o for y in o[b])
print("stft next to foreign s3 convs: bad runs of 40:", nbad)
# and the full failing pattern, checking the kept mags afterwards
keep = {}
def mel_keep2(xs):
    B = xs.shape[0]
    b = next(bb for bb in B3 if bb[1] - bb[0] == B and xs.data_ptr() == x[bb[0]:bb[1]].data_ptr())
    mag = stft_only(xs); keep[b] = mag
    mel = e.conv(mag, None, B * cfg.n_frames, 1, 1, e.ldm, 1, 1, cfg.n_mels, 1, 1, act=E.ACT_LOGCLAMP, act_param=cfg.log_floor, bias=False, w=e.w["mel_fb"], w3key="mel_fb")
    return mel.view(B, cfg.n_frames, cfg.n_mels, 1)
lref = {b: e.log_mel(x[b[0]:b[1]]).clone() for b in B3}
bad_mel = bad_mag = 0
for rep in range(40):
    o = run(mel_keep2)
    bad_mel += any(float((o[b] - lref[b]).abs().max()) > 0 for b in B3)
    bad_mag += any(float((keep[b] - mref[b]).abs().max()) > 0 for b in B3)
print("stft -> conv per stream: runs with wrong mel", bad_mel, " runs with wrong mag afterwards", bad_mag)
print("--- structure of wrong mag values")
shown = 0
for rep in range(40):
    o = run_mixed2()
    for b in o:
        for y in o[b]:
            d = (y - mref[b]).abs().view(-1, e.ldm)
            if float(d.max()) > 0:
                nz = (d > 0).nonzero().cpu()
                rows = nz[:, 0].unique().tolist()
                r = rows[0]
                cols = nz[nz[:, 0] == r][:, 1].tolist()
                g, w = y.view(-1, e.ldm)[r], mref[b].view(-1, e.ldm)[r]
                print("rows wrong", len(rows), rows[:6], "| row", r, "wrong cols", len(cols), cols[:12], "...", cols[-3:],
                      "| got", [round(float(g[c]), 4) for c in cols[:4]], "want", [round(float(w[c]), 4) for c in cols[:4]])
                shown += 1
                break
        if shown >= 5: break
    if shown >= 5: break
