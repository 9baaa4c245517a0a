"""`Egregora DeepFilterNet Denoise` (the first stage of BASELINE configs[4]) with the
reference's plugin surface (egregora_audio_enhance_extras.py:450-724, fixture G11).

What runs where: the DeepFilterNet network itself is upstream's PyTorch package (`df`), exactly as in the reference -- it
is absent from this build's image and is NOT re-implemented here; when `df` is not importable and no enhancer was
registered with `set_enhancer`, execute() raises.  Everything around the model -- 10 ms RMS VAD with its 95th-percentile
normalisation, one-pole smoothing, adaptive strength, equal-power / linear wet-dry gains, clip, post-gain, ceiling limiter,
and the 48 kHz rate conversion -- runs in libegregora_amd.so (egr_dfn_vad_gains, egr_dfn_mix, egr_resample_poly), with
float32 roundings placed exactly where numpy places them in the reference (oracle/dfn_mix.py reproduces G11 bit for bit).

Deliberate, documented differences from the reference:
  Q6  adaptive_vad_source = "none" (and "rnnoise" without pyrnnoise) makes the reference raise a broadcast ValueError for
      every input longer than 10 ms (:672-687); here the strength is simply constant.
  Q7  rate conversion to / from 48 kHz uses this pack's scipy-polyphase kernel instead of df.io.resample.
"""
from typing import Callable, Optional

import torch

from . import native, resample

_ENHANCER: Optional[Callable[[torch.Tensor, str], torch.Tensor]] = None


def set_enhancer(fn: Optional[Callable[[torch.Tensor, str], torch.Tensor]]):
    """Register the denoiser backend: fn(x48 [1,T] float32 CPU tensor, model_name) -> [1,T] tensor.  None restores the
    default (upstream `df.enhance`)."""
    global _ENHANCER
    _ENHANCER = fn


def _batched(wav: torch.Tensor, ranks, err: str) -> torch.Tensor:
    """Left-pad the shape with unit axes up to [B,C,T]; ranks outside `ranks` are refused with the reference's message."""
    if wav.dim() not in ranks:
        raise ValueError(err)
    return wav.reshape((1,) * (3 - wav.dim()) + tuple(wav.shape)).float()


def _coerce_audio(x):
    """-> (wave [B,C,T] float32, sr, meta).  Accepts what the reference's enhance nodes accept (:29-52): an AUDIO dict whose
    waveform has 1-3 axes, or a bare [C,T] / [B,C,T] tensor (taken as 48 kHz); same exception types and messages."""
    from .audio_glue import is_audio_dict
    if is_audio_dict(x):
        return (_batched(x["waveform"], (1, 2, 3), "Audio waveform must be 1D, 2D or 3D [B,C,T]."), int(x["sample_rate"]),
                x.get("meta", {}))
    if torch.is_tensor(x):
        return _batched(x, (2, 3), "Tensor audio must be [C,T] or [B,C,T]."), 48000, {}
    raise TypeError("Unsupported audio input type.")


def _make_audio(sr: int, wav: torch.Tensor, meta: Optional[dict] = None):
    """[C,T] / [B,C,T] -> the enhance pack's AUDIO dict (waveform contiguous [B,C,T], sample_rate, meta)."""
    if wav.dim() not in (2, 3):
        raise ValueError("samples must be 1D/2D/3D; got shape %r" % (wav.shape,))
    return dict(waveform=wav.reshape((1,) * (3 - wav.dim()) + tuple(wav.shape)).contiguous(), sample_rate=int(sr), meta=meta or {})


_MODES = {"off": 0, "more_on_noise": 1, "more_on_speech": 2, "gate_on_noise": 3}


def mix_on_device(dry: torch.Tensor, wet: torch.Tensor, dry48: Optional[torch.Tensor], sr: int, strength, mix_curve,
                  adaptive_mode, adaptive_amount, vad_threshold, vad_smooth_ms, post_gain_db, limit_ceiling, ceiling):
    """dry, wet [C,T] CUDA float32 at `sr`; dry48 the 48 kHz dry signal for the VAD (None: constant strength).  -> [C,T]."""
    import ctypes as C
    L = native.lib()
    Cn, T = dry.shape
    dev = dry.device
    if dry48 is not None:
        n48 = dry48.shape[1]
        nfr = (n48 + 479) // 480
        ws = torch.empty(int(L.egr_dfn_workspace_bytes(Cn, n48)), dtype=torch.uint8, device=dev)
        gd = torch.empty((Cn, nfr), dtype=torch.float32, device=dev)
        gw = torch.empty((Cn, nfr), dtype=torch.float32, device=dev)
        native.check(L.egr_dfn_vad_gains(native.ptr(dry48), Cn, n48, float(vad_smooth_ms), _MODES.get(adaptive_mode, 0),
                                         float(strength), float(adaptive_amount), float(vad_threshold),
                                         0 if mix_curve == "equal_power" else 1, native.ptr(ws), native.ptr(gd), native.ptr(gw),
                                         native.stream_ptr()), "egr_dfn_vad_gains")
        hop = max(1, int(sr * 0.010))
    else:           # no VAD: one gain pair for the whole signal (Q6)
        import math
        s = min(max(float(strength), 0.0), 1.0)
        g = (math.cos(0.5 * math.pi * s), math.sin(0.5 * math.pi * s)) if mix_curve == "equal_power" else (1.0 - s, s)
        gd = torch.full((Cn, 1), g[0], dtype=torch.float32, device=dev)
        gw = torch.full((Cn, 1), g[1], dtype=torch.float32, device=dev)
        nfr, hop = 1, max(1, T)
    y = torch.empty_like(dry)
    peak = torch.zeros(1, dtype=torch.int32, device=dev)
    gain = float(10.0 ** (post_gain_db / 20.0))
    native.check(L.egr_dfn_mix(native.ptr(dry), native.ptr(wet), native.ptr(gd), native.ptr(gw), Cn, T, nfr, hop, gain,
                               1 if post_gain_db != 0.0 else 0, 1 if limit_ceiling else 0, float(ceiling), native.ptr(y),
                               native.ptr(peak), native.stream_ptr()), "egr_dfn_mix")
    return y


class Egregora_DeepFilterNet_Denoise:
    @classmethod
    def INPUT_TYPES(cls):
        return {
            "required": {
                "audio": ("AUDIO",),
                "dfn_model": (["DeepFilterNet2", "DeepFilterNet3"], {"default": "DeepFilterNet2"}),
                "device": (["auto", "cuda:0", "cpu"], {"default": "auto"}),
                "use_postfilter": ("BOOLEAN", {"default": False, "label_on": "postfilter on", "label_off": "postfilter off"}),
                "limit_ceiling": ("BOOLEAN", {"default": True, "label_on": "limit on", "label_off": "limit off"}),
                "stereo_mode": (["per_channel", "downmix_mono"], {"default": "per_channel"}),
                "frame_ms": ("INT", {"default": 20, "min": 5, "max": 60, "step": 5}),
                "strength": ("FLOAT", {"default": 0.65, "min": 0.0, "max": 1.0, "step": 0.01}),
                "mix_curve": (["equal_power", "linear"], {"default": "equal_power"}),
                "adaptive_vad_source": (["rms", "rnnoise", "none"], {"default": "rms"}),
                "adaptive_mode": (["off", "more_on_noise", "more_on_speech", "gate_on_noise"], {"default": "more_on_noise"}),
                "adaptive_amount": ("FLOAT", {"default": 0.45, "min": 0.0, "max": 1.0, "step": 0.01}),
                "vad_threshold": ("FLOAT", {"default": 0.90, "min": 0.0, "max": 1.0, "step": 0.01}),
                "vad_smooth_ms": ("INT", {"default": 60, "min": 0, "max": 500, "step": 5}),
                "post_gain_db": ("FLOAT", {"default": 0.5, "min": -24.0, "max": 24.0, "step": 0.1}),
                "ceiling": ("FLOAT", {"default": 0.98, "min": 0.1, "max": 1.0, "step": 0.001}),
            }
        }

    RETURN_TYPES = ("AUDIO",)
    FUNCTION = "execute"
    CATEGORY = "Egregora/Enhance"

    _DF_CACHE = {}

    def _pick_device(self, choice: str):
        if choice == "auto":
            return "cuda:0" if torch.cuda.is_available() else "cpu"
        return choice

    def _enhance(self, x48_cpu: torch.Tensor, model_name: str, dev: str) -> torch.Tensor:
        """The denoiser proper: a registered backend, else upstream DeepFilterNet as the reference drives it (:509-517,636-647)."""
        if _ENHANCER is not None:
            return torch.cat([_ENHANCER(x48_cpu[c:c + 1], model_name) for c in range(x48_cpu.shape[0])], 0)
        try:
            from df.enhance import enhance, init_df
        except Exception as e:      # noqa: BLE001
            raise RuntimeError("DeepFilterNet (python package `df`) is not installed; this pack runs the stage around the "
                               "model on the GPU but does not re-implement the upstream network "
                               "(register one with egregora_audio_enhance_extras.set_enhancer).") from e
        key = (model_name, dev)
        if key not in self._DF_CACHE:
            model, df_state, _ = init_df(model_name, config_allow_defaults=True)
            self._DF_CACHE[key] = (model.to(dev).eval(), df_state)
        model, df_state = self._DF_CACHE[key]
        with torch.no_grad():
            return torch.cat([enhance(model, df_state, x48_cpu[c:c + 1]) for c in range(x48_cpu.shape[0])], 0)

    def execute(self, audio, dfn_model="DeepFilterNet2", device="auto", use_postfilter=False, limit_ceiling=True,
                stereo_mode="per_channel", frame_ms=20, strength=0.65, mix_curve="equal_power", adaptive_vad_source="rms",
                adaptive_mode="more_on_noise", adaptive_amount=0.45, vad_threshold=0.90, vad_smooth_ms=60, post_gain_db=0.5,
                ceiling=0.98):
        native.require_device()
        wav, sr, meta = _coerce_audio(audio)
        if stereo_mode == "downmix_mono" and wav.size(1) != 1:
            wav = wav.mean(dim=1, keepdim=True)
        B, C, T = wav.shape
        dry = wav.reshape(-1, T).to(torch.float32).contiguous().cuda()
        dry48 = resample.resample_hq(dry, sr, 48000) if sr != 48000 else dry
        dev = self._pick_device(device)
        wet48 = self._enhance(dry48.cpu(), dfn_model, dev).to(torch.float32).cuda().contiguous()
        wet = resample.resample_hq(wet48, 48000, sr) if sr != 48000 else wet48
        if wet.shape[1] != T:                      # polyphase lengths can differ by a sample after the round trip
            wet = torch.nn.functional.pad(wet, (0, max(0, T - wet.shape[1])))[:, :T].contiguous()
        vad_src = dry48.contiguous() if adaptive_vad_source == "rms" else None          # Q6: "none" / "rnnoise"
        y = mix_on_device(dry, wet, vad_src, sr, strength, mix_curve, adaptive_mode, adaptive_amount, vad_threshold,
                          vad_smooth_ms, post_gain_db, limit_ceiling, ceiling)
        meta2 = dict(meta)
        meta2["deepfilternet"] = {
            "model": dfn_model, "device": dev, "use_postfilter": bool(use_postfilter), "stereo_mode": stereo_mode,
            "frame_ms": frame_ms, "strength": strength, "mix_curve": mix_curve, "adaptive_vad_source": adaptive_vad_source,
            "adaptive_mode": adaptive_mode, "adaptive_amount": adaptive_amount, "vad_threshold": vad_threshold,
            "vad_smooth_ms": vad_smooth_ms, "post_gain_db": post_gain_db, "limit_ceiling": bool(limit_ceiling), "ceiling": ceiling,
        }
        return (_make_audio(sr, y.cpu().reshape(B, C, -1), meta2),)


NODE_CLASS_MAPPINGS = {"Egregora_DeepFilterNet_Denoise": Egregora_DeepFilterNet_Denoise}
NODE_DISPLAY_NAME_MAPPINGS = {"Egregora_DeepFilterNet_Denoise": "Egregora DeepFilterNet Denoise"}
