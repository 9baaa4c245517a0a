"""ORACLE (test infrastructure, not product): plain PyTorch fp32 implementation of the FlashSR layer table
declared in the product's flashsr_arch.py.  It is the build-internal reference each HIP kernel and the assembled
engine are compared against (torch fp32 ops of the same graph), NOT upstream FlashSR.

PARITY UNPINNED vs upstream: `FlashSR_Inference` and its checkpoints are absent from the reference tree and the
image (reference egregora_audio_super_resolution.py:65-68,260-261,323); see flashsr_arch.py for what is recalled.
Call contract anchored on the reference: model(x[B,245760]) -> y[B,245760] at 48 kHz (:366-369).
"""
import math

import torch
import torch.nn.functional as F


def _gn(x, P, name, groups, eps):
    return F.group_norm(x, groups, P[name + ".weight"], P[name + ".bias"], eps)


def _conv2(x, P, name, stride=1, pad=1):
    return F.conv2d(x, P[name + ".weight"], P[name + ".bias"], stride=stride, padding=pad)


# ------------------------------------------------------------------ mel front-end
def log_mel(x, cfg, mel_fb):
    """x [B,L] -> [B,1,n_frames,n_mels]: reflect pad (n_fft-hop)/2, periodic Hann, |STFT|, mel, log(clamp)."""
    p = (cfg.n_fft - cfg.hop) // 2
    xp = F.pad(x[:, None, :], (p, p), mode="reflect")[:, 0]
    win = torch.hann_window(cfg.n_fft, periodic=True, dtype=x.dtype, device=x.device)
    fr = xp.unfold(1, cfg.n_fft, cfg.hop) * win                       # [B,T,n_fft]
    mag = torch.fft.rfft(fr, dim=-1).abs()                             # [B,T,nb]
    mel = mag @ mel_fb.t()                                             # [B,T,n_mels]
    T = mel.shape[1]
    if T < cfg.n_frames:
        mel = F.pad(mel, (0, 0, 0, cfg.n_frames - T))
    return torch.log(torch.clamp(mel[:, :cfg.n_frames], min=cfg.log_floor))[:, None]


def lowpass_ref(x, cfg, pct=0.985, order=8, ripple_db=0.05):
    """lowpass_input=True as this build defines it: per row, cutoff bin = highest STFT bin whose cumulative
    (time-summed) magnitude is still below pct of the total, + 1 (the scan of AudioSR's `_find_cutoff`, as recalled);
    zero-phase Chebyshev-I amplitude gain 1/(1 + eps^2 T_n^2(tan(pi f/sr)/tan(pi fc/sr))) applied to rfft bins."""
    p = (cfg.n_fft - cfg.hop) // 2
    xp = F.pad(x[:, None, :], (p, p), mode="reflect")[:, 0]
    win = torch.hann_window(cfg.n_fft, periodic=True, dtype=x.dtype)
    mag = torch.fft.rfft(xp.unfold(1, cfg.n_fft, cfg.hop) * win, dim=-1).abs()      # [B,T,nb]
    e = mag.sum(1).double()
    nb = e.shape[1]
    ys, cuts = [], []
    L = x.shape[1]
    nbins = L // 2 + 1
    f = 0.5 * cfg.sr * torch.arange(nbins, dtype=torch.float64) / (nbins - 1)
    eps2 = 10.0 ** (ripple_db / 10.0) - 1.0
    for b in range(x.shape[0]):
        c = torch.cumsum(e[b], 0)
        lim = c[-1] * pct
        cut = 0
        for i in range(1, nb):
            if c[nb - i] < lim:
                cut = nb - i
                break
        cuts.append(cut)
        fc = max(1.0, min(cut / (nb - 1), 0.999) * 0.5 * cfg.sr)
        wc = math.tan(math.pi * fc / cfg.sr)
        xw = torch.tan(math.pi * torch.clamp(f, max=0.4999 * cfg.sr) / cfg.sr) / wc
        tn = torch.where(xw <= 1, torch.cos(order * torch.acos(torch.clamp(xw, max=1.0))),
                         torch.cosh(order * torch.acosh(torch.clamp(xw, min=1.0))))
        g = 1.0 / (1.0 + eps2 * tn * tn)
        g = torch.where(f < 0.4999 * cfg.sr, g, torch.zeros_like(g))
        ys.append(torch.fft.irfft(torch.fft.rfft(x[b].double()) * g, n=L).to(x.dtype))
    return torch.stack(ys), cuts


# ------------------------------------------------------------------ VAE
def _vae_res(x, P, name, G):
    h = _conv2(F.silu(_gn(x, P, name + ".norm1", G, 1e-6)), P, name + ".conv1")
    h = _conv2(F.silu(_gn(h, P, name + ".norm2", G, 1e-6)), P, name + ".conv2")
    if name + ".nin_shortcut.weight" in P:
        x = _conv2(x, P, name + ".nin_shortcut", pad=0)
    return x + h


def _vae_attn(x, P, name, G):
    B, C, H, W = x.shape
    h = _gn(x, P, name + ".norm", G, 1e-6)
    q = _conv2(h, P, name + ".q", pad=0).reshape(B, C, H * W).permute(0, 2, 1)
    k = _conv2(h, P, name + ".k", pad=0).reshape(B, C, H * W)
    v = _conv2(h, P, name + ".v", pad=0).reshape(B, C, H * W)
    w = torch.softmax(torch.bmm(q, k) * (C ** -0.5), dim=2)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(B, C, H, W)
    return x + _conv2(h, P, name + ".proj_out", pad=0)


def vae_encode(mel, P, cfg):
    G = cfg.gn_groups
    h = _conv2(mel, P, "vae.encoder.conv_in")
    n = len(cfg.vae_mult)
    for lv in range(n):
        for b in range(cfg.vae_res):
            h = _vae_res(h, P, f"vae.encoder.down.{lv}.block.{b}", G)
        if lv != n - 1:
            h = _conv2(F.pad(h, (0, 1, 0, 1)), P, f"vae.encoder.down.{lv}.downsample.conv", stride=2, pad=0)
    h = _vae_res(h, P, "vae.encoder.mid.block_1", G)
    h = _vae_attn(h, P, "vae.encoder.mid.attn_1", G)
    h = _vae_res(h, P, "vae.encoder.mid.block_2", G)
    h = _conv2(F.silu(_gn(h, P, "vae.encoder.norm_out", G, 1e-6)), P, "vae.encoder.conv_out")
    moments = _conv2(h, P, "vae.quant_conv", pad=0)
    return moments[:, :cfg.z_ch]                                       # posterior mean


def vae_decode(z, P, cfg):
    G = cfg.gn_groups
    h = _conv2(z, P, "vae.post_quant_conv", pad=0)
    h = _conv2(h, P, "vae.decoder.conv_in")
    h = _vae_res(h, P, "vae.decoder.mid.block_1", G)
    h = _vae_attn(h, P, "vae.decoder.mid.attn_1", G)
    h = _vae_res(h, P, "vae.decoder.mid.block_2", G)
    for lv in reversed(range(len(cfg.vae_mult))):
        for b in range(cfg.vae_res + 1):
            h = _vae_res(h, P, f"vae.decoder.up.{lv}.block.{b}", G)
        if lv != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv2(h, P, f"vae.decoder.up.{lv}.upsample.conv")
    return _conv2(F.silu(_gn(h, P, "vae.decoder.norm_out", G, 1e-6)), P, "vae.decoder.conv_out")


# ------------------------------------------------------------------ UNet
def timestep_embedding(t, dim, device, dtype=torch.float32):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=dtype, device=device) / half)
    args = torch.tensor([float(t)], device=device, dtype=dtype)[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)      # [1,dim]


def _attn_tokens(x, P, name, head_dim):
    B, T, C = x.shape
    H = C // head_dim
    q = F.linear(x, P[name + ".to_q.weight"]).reshape(B, T, H, head_dim).permute(0, 2, 1, 3)
    k = F.linear(x, P[name + ".to_k.weight"]).reshape(B, T, H, head_dim).permute(0, 2, 1, 3)
    v = F.linear(x, P[name + ".to_v.weight"]).reshape(B, T, H, head_dim).permute(0, 2, 1, 3)
    w = torch.softmax(q @ k.transpose(-1, -2) * (head_dim ** -0.5), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    return F.linear(o, P[name + ".to_out.weight"], P[name + ".to_out.bias"])


def _unet_block(x, temb_act, P, base, cfg, has_attn):
    G = cfg.gn_groups
    h = _conv2(F.silu(_gn(x, P, base + ".res.in_norm", G, 1e-5)), P, base + ".res.in_conv")
    e = F.linear(temb_act, P[base + ".res.emb.weight"], P[base + ".res.emb.bias"])
    h = h + e[:, :, None, None]
    h = _conv2(F.silu(_gn(h, P, base + ".res.out_norm", G, 1e-5)), P, base + ".res.out_conv")
    if base + ".res.skip.weight" in P:
        x = _conv2(x, P, base + ".res.skip", pad=0)
    x = x + h
    if has_attn:
        B, C, H, W = x.shape
        x_in = x
        t = _conv2(_gn(x, P, base + ".st.norm", G, 1e-6), P, base + ".st.proj_in", pad=0)
        t = t.reshape(B, C, H * W).permute(0, 2, 1)
        for a in ("attn1", "attn2"):
            t = t + _attn_tokens(F.layer_norm(t, (C,), P[f"{base}.st.{a}_ln.weight"], P[f"{base}.st.{a}_ln.bias"]),
                                 P, f"{base}.st.{a}", cfg.head_dim)
        u = F.layer_norm(t, (C,), P[base + ".st.ff_ln.weight"], P[base + ".st.ff_ln.bias"])
        u = F.linear(u, P[base + ".st.ff.geglu.weight"], P[base + ".st.ff.geglu.bias"])
        a_, gate = u.chunk(2, dim=-1)
        u = F.linear(a_ * F.gelu(gate), P[base + ".st.ff.out.weight"], P[base + ".st.ff.out.bias"])
        t = t + u
        t = t.permute(0, 2, 1).reshape(B, C, H, W)
        x = x_in + _conv2(t, P, base + ".st.proj_out", pad=0)
    return x


def unet(x, t, P, cfg, blocks):
    temb = timestep_embedding(t, cfg.unet_ch, x.device, x.dtype)
    temb = F.linear(temb, P["unet.time_embed.0.weight"], P["unet.time_embed.0.bias"])
    temb = F.linear(F.silu(temb), P["unet.time_embed.2.weight"], P["unet.time_embed.2.bias"])
    temb_act = F.silu(temb)
    skips = []
    h = x
    for name, cin, cout, attn in blocks:
        part, _, kind = name.split(".")
        base = f"unet.{name}"
        if kind == "conv_in":
            h = _conv2(h, P, base)
            skips.append(h)
        elif kind == "down":
            h = _conv2(h, P, base + ".conv", stride=2, pad=1)
            skips.append(h)
        elif kind == "up":
            h = _conv2(F.interpolate(h, scale_factor=2.0, mode="nearest"), P, base + ".conv")
        else:
            if part == "out":
                h = torch.cat([h, skips.pop()], dim=1)
            h = _unet_block(h, temb_act, P, base, cfg, attn)
            if part == "in":
                skips.append(h)
    return _conv2(F.silu(_gn(h, P, "unet.out_norm", cfg.gn_groups, 1e-5)), P, "unet.out_conv")


# ------------------------------------------------------------------ vocoder
def _snake(x, alpha, beta):
    a = torch.exp(alpha)[None, :, None]
    b = torch.exp(beta)[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def _act_aa(x, alpha, beta, filt):
    """Anti-aliased snake: 2x up (zero-insert + kaiser-sinc FIR, replicate pad), snake, 2x down (FIR, stride 2)."""
    B, C, L = x.shape
    k = filt.shape[0]
    f = filt[None, None, :].expand(C, 1, k)
    pad = k // 2 - 1
    pl = pad * 2 + (k - 2) // 2
    pr = pad * 2 + (k - 2 + 1) // 2
    u = F.pad(x, (pad, pad), mode="replicate")
    u = 2.0 * F.conv_transpose1d(u, f, stride=2, groups=C)
    u = u[..., pl:-pr]
    u = _snake(u, alpha, beta)
    d = F.pad(u, (k // 2 - 1, k // 2), mode="replicate")
    return F.conv1d(d, f, stride=2, groups=C)


def vocoder(mel, wave, P, cfg, filt):
    """mel [B,1,T,F] (log-mel image), wave [B,L] (the LR input) -> [B,L]."""
    n = len(cfg.voc_rates)
    m = mel[:, 0].permute(0, 2, 1)                                    # [B,F,T]
    h = F.conv1d(m, P["voc.conv_pre.weight"], P["voc.conv_pre.bias"], padding=3)
    feats = []
    e = wave[:, None, :]
    for i, r in enumerate(reversed(cfg.voc_rates)):
        e = F.leaky_relu(F.conv1d(e, P[f"voc.wave_enc.{i}.weight"], P[f"voc.wave_enc.{i}.bias"], stride=r, padding=r), 0.1)
        feats.append(e)
    h = h + feats[n - 1]
    for j, r in enumerate(cfg.voc_rates):
        kt = 2 * r + (r % 2)
        h = F.conv_transpose1d(h, P[f"voc.ups.{j}.weight"], P[f"voc.ups.{j}.bias"], stride=r, padding=(kt - r) // 2)
        if j <= n - 2:
            h = h + feats[n - 2 - j]
        acc = None
        for ki, k in enumerate(cfg.voc_kernels):
            x = h
            for di, d in enumerate(cfg.voc_dils):
                b = f"voc.amp.{j}.{ki}.{di}"
                xt = _act_aa(x, P[b + ".alpha1"], P[b + ".beta1"], filt)
                xt = F.conv1d(xt, P[b + ".conv1.weight"], P[b + ".conv1.bias"], dilation=d, padding=d * (k - 1) // 2)
                xt = _act_aa(xt, P[b + ".alpha2"], P[b + ".beta2"], filt)
                xt = F.conv1d(xt, P[b + ".conv2.weight"], P[b + ".conv2.bias"], padding=(k - 1) // 2)
                x = xt + x
            acc = x if acc is None else acc + x
        h = acc / len(cfg.voc_kernels)
    h = _act_aa(h, P["voc.post.alpha"], P["voc.post.beta"], filt)
    h = F.conv1d(h, P["voc.conv_post.weight"], P["voc.conv_post.bias"], padding=3)
    return torch.tanh(h)[:, 0]


# ------------------------------------------------------------------ whole model
def to_float64(P):
    """The same parameters as float64 tensors: flashsr_forward then runs every op in double precision -- the yardstick
    that separates the fp32 round-off of EITHER implementation from a real disagreement (tests/test_gpu_flashsr.py)."""
    return {k: v.double() for k, v in P.items()}


def flashsr_forward(x, noise, P, cfg, blocks, mel_fb, filt, stages=None):
    """x [B,chunk], noise [B,z_ch,h,w] -> y [B,chunk], in the dtype of the arguments (float32: the oracle; float64: the
    yardstick).  `stages` (dict) receives the intermediates."""
    mel = log_mel(x, cfg, mel_fb)
    z_c = vae_encode(mel, P, cfg)
    t = cfg.t_steps - 1
    s = 0.008
    f = lambda u: math.cos((u / cfg.t_steps + s) / (1 + s) * math.pi / 2) ** 2
    abar = min(max(f(t + 1) / f(0), 1e-5), 0.99999)
    alpha, sigma = math.sqrt(abar), math.sqrt(1.0 - abar)
    v = unet(torch.cat([noise, z_c], dim=1), t, P, cfg, blocks)
    z0 = alpha * noise - sigma * v
    mel_hat = vae_decode(z0, P, cfg)
    y = vocoder(mel_hat, x, P, cfg, filt)
    if stages is not None:
        stages.update(mel=mel, z_cond=z_c, v=v, z0=z0, mel_hat=mel_hat, y=y)
    return y[:, :x.shape[1]]
