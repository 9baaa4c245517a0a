"""Declared layer table of the FlashSR network as this build executes it.

PARITY UNPINNED.  The reference pack delegates the model to `FlashSR.FlashSR.FlashSR` from the un-pinned
GitHub zip `jakeoneijk/FlashSR_Inference@main` plus three checkpoints from an HF dataset
(reference egregora_audio_super_resolution.py:65-68, 260-261, 323, 353); neither the code nor the weights exist
in the reference tree or in the build image.  What the reference does pin is the call contract
(`model(x[C,245760] @48 kHz, lowpass_input=bool) -> y[C, >=245760]`, :366-369) and the stage list named by
BASELINE.json (student_ldm 1-step UNet, VAE, sr_vocoder, mel/STFT front-end).

The table below follows the published AudioSR / FlashSR design as recalled (UPSTREAM-RECALL):
  mel front-end   : STFT 2048/480 Hann, 256 slaney mel bands 20 Hz..24 kHz, log(clamp(.,1e-5))   -> [1,512,256]
  VAE encoder     : AutoencoderKL, ch 128, mult (1,2,4,8), 2 res-blocks/level, z = 16 ch          -> [16,64,32]
  student UNet    : in 32 (noise ++ cond latent), model ch 128, mult (1,2,3,5), 2 res-blocks/level,
                    spatial transformers at ds 2/4/8, head dim 32, one v-prediction step at t = T-1
  VAE decoder     : mirror of the encoder with 3 res-blocks/level + mid attention                -> [1,512,256]
  SR vocoder      : BigVGAN-style generator (snake + anti-aliased resampling AMP blocks, transposed-conv
                    up-rates 6*5*4*2*2 = 480) with a strided-conv encoder of the LR waveform added U-Net style
Every dimension is a field of `FlashSRConfig`; `config_from_params` re-derives the widths from checkpoint tensor
shapes when real weights are supplied, and `init_params` produces seeded synthetic weights of the same shapes
for benchmarking (BASELINE: "random-init weights of that architecture").  The same table drives the HIP engine
(flashsr_engine.py), the PyTorch fp32 reference (oracle/flashsr_torch.py) and the FLOP count used for the MFMA
roofline (FlashSREngine.flop_count walks the executed graph).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass
class FlashSRConfig:
    sr: int = 48000
    chunk: int = 245760
    n_fft: int = 2048
    hop: int = 480
    n_mels: int = 256
    fmin: float = 20.0
    fmax: float = 24000.0
    n_frames: int = 512
    log_floor: float = 1e-5
    # VAE
    vae_ch: int = 128
    vae_mult: Tuple[int, ...] = (1, 2, 4, 8)
    vae_res: int = 2
    z_ch: int = 16
    gn_groups: int = 32
    # UNet
    unet_ch: int = 128
    unet_mult: Tuple[int, ...] = (1, 2, 3, 5)
    unet_res: int = 2
    unet_attn_ds: Tuple[int, ...] = (2, 4, 8)
    head_dim: int = 32
    # diffusion (cosine schedule, v-prediction, one step from t = T-1)
    t_steps: int = 1000
    # vocoder
    voc_ch: int = 512
    voc_rates: Tuple[int, ...] = (6, 5, 4, 2, 2)
    voc_kernels: Tuple[int, ...] = (3, 7, 11)
    voc_dils: Tuple[int, ...] = (1, 3, 5)
    aa_taps: int = 12

    @property
    def lat_hw(self):
        d = 2 ** (len(self.vae_mult) - 1)
        return self.n_frames // d, self.n_mels // d


TINY = dict(chunk=3840, n_fft=128, hop=30, n_mels=32, n_frames=128, fmax=24000.0,
            vae_ch=32, vae_mult=(1, 2), vae_res=1, z_ch=4, gn_groups=8,
            unet_ch=32, unet_mult=(1, 2), unet_res=1, unet_attn_ds=(2,), head_dim=16,
            voc_ch=32, voc_rates=(5, 3, 2), voc_kernels=(3, 5), voc_dils=(1, 3))


def tiny_config() -> FlashSRConfig:
    """Same topology at toy sizes (3840-sample chunk) for CPU-side tests of the reference graph."""
    return FlashSRConfig(**TINY)


# ---------------------------------------------------------------------------------------------
# mel filterbank (slaney scale + slaney norm, as librosa.filters.mel(htk=False, norm='slaney'))
# ---------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(cfg: FlashSRConfig) -> np.ndarray:
    """[n_mels, n_fft/2+1] float32."""
    nb = cfg.n_fft // 2 + 1
    fft_f = np.linspace(0.0, cfg.sr / 2.0, nb)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(cfg.fmin), _hz_to_mel(cfg.fmax), cfg.n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((cfg.n_mels, nb))
    for i in range(cfg.n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:cfg.n_mels + 2] - mel_f[:cfg.n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def kaiser_sinc_filter(taps: int, cutoff: float = 0.25, half_width: float = 0.3) -> np.ndarray:
    """Low-pass used by the anti-aliased activations (BigVGAN alias-free-torch `kaiser_sinc_filter1d`)."""
    even = taps % 2 == 0
    half = taps // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21) if A >= 21 else 0.0)
    win = np.kaiser(taps, beta)
    t = (np.arange(-half, half) + 0.5) if even else (np.arange(taps) - half)
    f = 2 * cutoff * win * np.sinc(2 * cutoff * t)
    return (f / f.sum()).astype(np.float32)


def cosine_alpha_sigma(cfg: FlashSRConfig, t: int):
    """alpha_t, sigma_t of the cosine schedule (Nichol & Dhariwal), alpha^2 + sigma^2 = 1."""
    s = 0.008
    f = lambda u: math.cos((u / cfg.t_steps + s) / (1 + s) * math.pi / 2) ** 2
    abar = min(max(f(t + 1) / f(0), 1e-5), 0.99999)
    return math.sqrt(abar), math.sqrt(1.0 - abar)


# ---------------------------------------------------------------------------------------------
# parameter table
# ---------------------------------------------------------------------------------------------
def _conv(P, name, cin, cout, kh, kw, gain=1.0, g=None):
    fan = cin * kh * kw
    P[name + ".weight"] = (torch.randn(cout, cin, kh, kw, generator=g) * (gain / math.sqrt(fan)))
    P[name + ".bias"] = torch.randn(cout, generator=g) * 0.02


def _lin(P, name, cin, cout, gain=1.0, g=None, bias=True):
    P[name + ".weight"] = torch.randn(cout, cin, generator=g) * (gain / math.sqrt(cin))
    if bias:
        P[name + ".bias"] = torch.randn(cout, generator=g) * 0.02


def _norm(P, name, c, g=None):
    P[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
    P[name + ".bias"] = 0.05 * torch.randn(c, generator=g)


def _vae_res(P, name, cin, cout, g):
    _norm(P, name + ".norm1", cin, g)
    _conv(P, name + ".conv1", cin, cout, 3, 3, 1.0, g)
    _norm(P, name + ".norm2", cout, g)
    _conv(P, name + ".conv2", cout, cout, 3, 3, 0.5, g)
    if cin != cout:
        _conv(P, name + ".nin_shortcut", cin, cout, 1, 1, 1.0, g)


def _vae_attn(P, name, c, g):
    _norm(P, name + ".norm", c, g)
    for k in ("q", "k", "v"):
        _conv(P, f"{name}.{k}", c, c, 1, 1, 1.0, g)
    _conv(P, name + ".proj_out", c, c, 1, 1, 0.5, g)


def init_params(cfg: FlashSRConfig, seed: int = 0, shapes_only: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic float32 weights for every layer of the table (torch layouts: conv [Co,Ci,kh,kw]).
    shapes_only: the same table on the meta device (names and shapes, no storage) -- what checkpoint validation compares with."""
    if shapes_only:
        with torch.device("meta"):
            return init_params(cfg, seed)
    g = torch.Generator().manual_seed(seed) if torch.empty(0).device.type != "meta" else None
    P: Dict[str, torch.Tensor] = {}
    ch, mult = cfg.vae_ch, cfg.vae_mult
    # ---- VAE encoder ----
    _conv(P, "vae.encoder.conv_in", 1, ch, 3, 3, 1.0, g)
    cin = ch
    for lv, m in enumerate(mult):
        cout = ch * m
        for b in range(cfg.vae_res):
            _vae_res(P, f"vae.encoder.down.{lv}.block.{b}", cin, cout, g)
            cin = cout
        if lv != len(mult) - 1:
            _conv(P, f"vae.encoder.down.{lv}.downsample.conv", cin, cin, 3, 3, 1.0, g)
    _vae_res(P, "vae.encoder.mid.block_1", cin, cin, g)
    _vae_attn(P, "vae.encoder.mid.attn_1", cin, g)
    _vae_res(P, "vae.encoder.mid.block_2", cin, cin, g)
    _norm(P, "vae.encoder.norm_out", cin, g)
    _conv(P, "vae.encoder.conv_out", cin, 2 * cfg.z_ch, 3, 3, 1.0, g)
    _conv(P, "vae.quant_conv", 2 * cfg.z_ch, 2 * cfg.z_ch, 1, 1, 1.0, g)
    # ---- VAE decoder ----
    _conv(P, "vae.post_quant_conv", cfg.z_ch, cfg.z_ch, 1, 1, 1.0, g)
    cin = ch * mult[-1]
    _conv(P, "vae.decoder.conv_in", cfg.z_ch, cin, 3, 3, 1.0, g)
    _vae_res(P, "vae.decoder.mid.block_1", cin, cin, g)
    _vae_attn(P, "vae.decoder.mid.attn_1", cin, g)
    _vae_res(P, "vae.decoder.mid.block_2", cin, cin, g)
    for lv in reversed(range(len(mult))):
        cout = ch * mult[lv]
        for b in range(cfg.vae_res + 1):
            _vae_res(P, f"vae.decoder.up.{lv}.block.{b}", cin, cout, g)
            cin = cout
        if lv != 0:
            _conv(P, f"vae.decoder.up.{lv}.upsample.conv", cin, cin, 3, 3, 1.0, g)
    _norm(P, "vae.decoder.norm_out", cin, g)
    _conv(P, "vae.decoder.conv_out", cin, 1, 3, 3, 1.0, g)
    # ---- UNet ----
    mc, temb = cfg.unet_ch, 4 * cfg.unet_ch
    _lin(P, "unet.time_embed.0", mc, temb, 1.0, g)
    _lin(P, "unet.time_embed.2", temb, temb, 1.0, g)
    for name, cin_, cout_, attn in unet_blocks(cfg):
        kind = name.split(".")[-1]
        if kind == "conv_in":
            _conv(P, f"unet.{name}", cin_, cout_, 3, 3, 1.0, g)
        elif kind in ("down", "up"):
            _conv(P, f"unet.{name}.conv", cin_, cout_, 3, 3, 1.0, g)
        else:
            base = f"unet.{name}"
            _norm(P, base + ".res.in_norm", cin_, g)
            _conv(P, base + ".res.in_conv", cin_, cout_, 3, 3, 1.0, g)
            _lin(P, base + ".res.emb", temb, cout_, 0.3, g)
            _norm(P, base + ".res.out_norm", cout_, g)
            _conv(P, base + ".res.out_conv", cout_, cout_, 3, 3, 0.5, g)
            if cin_ != cout_:
                _conv(P, base + ".res.skip", cin_, cout_, 1, 1, 1.0, g)
            if attn:
                c = cout_
                _norm(P, base + ".st.norm", c, g)
                _conv(P, base + ".st.proj_in", c, c, 1, 1, 1.0, g)
                for a in ("attn1", "attn2"):
                    _norm(P, f"{base}.st.{a}_ln", c, g)
                    for k in ("to_q", "to_k", "to_v"):
                        _lin(P, f"{base}.st.{a}.{k}", c, c, 1.0, g, bias=False)
                    _lin(P, f"{base}.st.{a}.to_out", c, c, 0.5, g)
                _norm(P, base + ".st.ff_ln", c, g)
                _lin(P, base + ".st.ff.geglu", c, 8 * c, 1.0, g)
                _lin(P, base + ".st.ff.out", 4 * c, c, 0.5, g)
                _conv(P, base + ".st.proj_out", c, c, 1, 1, 0.5, g)
    _norm(P, "unet.out_norm", mc, g)
    _conv(P, "unet.out_conv", mc, cfg.z_ch, 3, 3, 1.0, g)
    # ---- SR vocoder ----
    vc = cfg.voc_ch
    P["voc.conv_pre.weight"] = torch.randn(vc, cfg.n_mels, 7, generator=g) / math.sqrt(cfg.n_mels * 7)
    P["voc.conv_pre.bias"] = torch.randn(vc, generator=g) * 0.02
    # LR-waveform encoder: strided convs mirroring the up-rates (finest first).  Feature e_i has length
    # L / prod(rates[n-1-i:]); e_{n-1} (mel-frame rate) is added to conv_pre's output, e_{n-2-j} to the output of
    # up stage j.
    n_up = len(cfg.voc_rates)
    chans = [vc // (2 ** (i + 1)) for i in range(n_up)]                         # channels after each up stage
    c_prev = 1
    for i, r in enumerate(reversed(cfg.voc_rates)):
        c_out = chans[n_up - 2 - i] if i <= n_up - 2 else vc
        P[f"voc.wave_enc.{i}.weight"] = torch.randn(c_out, c_prev, 2 * r + 1, generator=g) / math.sqrt(c_prev * (2 * r + 1))
        P[f"voc.wave_enc.{i}.bias"] = torch.randn(c_out, generator=g) * 0.02
        c_prev = c_out
    c = vc
    for i, r in enumerate(cfg.voc_rates):
        co = chans[i]
        kt = up_kernel(r)
        P[f"voc.ups.{i}.weight"] = torch.randn(c, co, kt, generator=g) / math.sqrt(c * 2)      # ConvTranspose1d [Ci,Co,k]
        P[f"voc.ups.{i}.bias"] = torch.randn(co, generator=g) * 0.02
        for j, k in enumerate(cfg.voc_kernels):
            for d_i, _ in enumerate(cfg.voc_dils):
                base = f"voc.amp.{i}.{j}.{d_i}"
                P[base + ".alpha1"] = 0.2 * torch.randn(co, generator=g)          # log-scale snake parameters
                P[base + ".beta1"] = 0.2 * torch.randn(co, generator=g)
                P[base + ".conv1.weight"] = torch.randn(co, co, k, generator=g) / math.sqrt(co * k)
                P[base + ".conv1.bias"] = torch.randn(co, generator=g) * 0.02
                P[base + ".alpha2"] = 0.2 * torch.randn(co, generator=g)
                P[base + ".beta2"] = 0.2 * torch.randn(co, generator=g)
                P[base + ".conv2.weight"] = 0.5 * torch.randn(co, co, k, generator=g) / math.sqrt(co * k)
                P[base + ".conv2.bias"] = torch.randn(co, generator=g) * 0.02
        c = co
    P["voc.post.alpha"] = 0.2 * torch.randn(c, generator=g)
    P["voc.post.beta"] = 0.2 * torch.randn(c, generator=g)
    P["voc.conv_post.weight"] = torch.randn(1, c, 7, generator=g) / math.sqrt(c * 7)
    P["voc.conv_post.bias"] = torch.zeros(1)
    return {k: v.float().contiguous() for k, v in P.items()}


def up_kernel(r: int) -> int:
    """Transposed-conv kernel for up-rate r: 2r (even r) or 2r+1 (odd r), padding (k-r)/2 => exact x r length."""
    return 2 * r + (r % 2)


def unet_blocks(cfg: FlashSRConfig) -> List[Tuple[str, int, int, bool]]:
    """Ordered (name, cin, cout, has_transformer) list of the UNet's blocks (openai-style topology:
    input blocks with skip stack, middle, output blocks consuming the skips)."""
    mc = cfg.unet_ch
    blocks = [("in.0.conv_in", 2 * cfg.z_ch, mc, False)]
    skip_ch = [mc]
    ch, ds = mc, 1
    idx = 1
    for lv, m in enumerate(cfg.unet_mult):
        for _ in range(cfg.unet_res):
            blocks.append((f"in.{idx}.block", ch, mc * m, ds in cfg.unet_attn_ds))
            ch = mc * m
            skip_ch.append(ch)
            idx += 1
        if lv != len(cfg.unet_mult) - 1:
            blocks.append((f"in.{idx}.down", ch, ch, False))
            skip_ch.append(ch)
            ds *= 2
            idx += 1
    blocks.append(("mid.0.block", ch, ch, True))
    blocks.append(("mid.1.block", ch, ch, False))
    idx = 0
    for lv in reversed(range(len(cfg.unet_mult))):
        m = cfg.unet_mult[lv]
        for i in range(cfg.unet_res + 1):
            sc = skip_ch.pop()
            blocks.append((f"out.{idx}.block", ch + sc, mc * m, ds in cfg.unet_attn_ds))
            ch = mc * m
            if lv != 0 and i == cfg.unet_res:
                blocks.append((f"out.{idx}.up", ch, ch, False))
                ds //= 2
            idx += 1
    return blocks


def config_from_params(P: Dict[str, torch.Tensor], base: FlashSRConfig = None) -> FlashSRConfig:
    """Re-derive the width / depth fields of the table from the tensor shapes of a state dict in this module's names (what
    flashsr_weights.map_checkpoints produces from the upstream files, or what `EGREGORA_FLASHSR_WEIGHTS` supplies).  Fields that
    shapes cannot reveal (sample rate, hop, schedule length, dilation values, GroupNorm groups, attention head width) keep the
    values of `base`; flashsr_weights then checks EVERY tensor shape of the resulting table against the checkpoint."""
    import dataclasses
    base = base or FlashSRConfig()
    idx = lambda prefix, pos: sorted({int(k.split(".")[pos]) for k in P if k.startswith(prefix)})
    vae_ch = P["vae.encoder.conv_in.weight"].shape[0]
    z_ch = P["vae.post_quant_conv.weight"].shape[0]
    levels = 1 + max(idx("vae.encoder.down.", 3))
    vae_mult = tuple(P[f"vae.encoder.down.{lv}.block.0.conv1.weight"].shape[0] // vae_ch for lv in range(levels))
    vae_res = 1 + max(idx("vae.encoder.down.0.block.", 5))
    # UNet: walk the input blocks; a `.down` entry closes a resolution level
    unet_ch = P["unet.time_embed.0.weight"].shape[1]
    mult, attn_ds, per_level, ds, n_here = [], [], [], 1, 0
    for i in idx("unet.in.", 2):
        if f"unet.in.{i}.block.res.in_conv.weight" in P:
            cout = P[f"unet.in.{i}.block.res.in_conv.weight"].shape[0]
            n_here += 1
            if f"unet.in.{i}.block.st.norm.weight" in P and ds not in attn_ds:
                attn_ds.append(ds)
            last = cout
        elif f"unet.in.{i}.down.conv.weight" in P:
            mult.append(last // unet_ch)
            per_level.append(n_here)
            n_here, ds = 0, ds * 2
    if n_here:
        mult.append(last // unet_ch)
        per_level.append(n_here)
    voc_ch = P["voc.conv_pre.weight"].shape[0]
    n_mels = P["voc.conv_pre.weight"].shape[1]
    rates = tuple(P[f"voc.ups.{i}.weight"].shape[2] // 2 for i in idx("voc.ups.", 2))          # kernel = 2r (+1 for odd r)
    kernels = tuple(P[f"voc.amp.0.{j}.0.conv1.weight"].shape[2] for j in idx("voc.amp.0.", 3))
    n_dil = 1 + max(idx("voc.amp.0.0.", 4))
    dils = base.voc_dils if len(base.voc_dils) == n_dil else tuple(2 * d + 1 for d in range(n_dil))
    lat_down = 2 ** (levels - 1)
    return dataclasses.replace(base, vae_ch=vae_ch, z_ch=z_ch, vae_mult=vae_mult, vae_res=vae_res, unet_ch=unet_ch,
                               unet_mult=tuple(mult) or base.unet_mult, unet_res=(per_level[0] if per_level else base.unet_res),
                               unet_attn_ds=tuple(attn_ds), voc_ch=voc_ch, n_mels=n_mels, voc_rates=rates, voc_kernels=kernels,
                               voc_dils=dils, hop=int(math.prod(rates)) if rates else base.hop,
                               n_frames=base.n_frames if (base.n_frames % lat_down == 0) else base.n_frames)
