#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE's own
host-side code from /root/reference (read-only) in the build container.

Only inputs/outputs (data) are written; no reference source text is stored.  The reference
cannot travel to the GPU box, so these fixtures are what pins the oracle there.

Fixture ids follow SURVEY.md section 8(c): G1..G9.

  python tests/golden/make_golden.py            # writes tests/golden/*.npz|*.json

Reference call sites exercised:
  egregora_audio_super_resolution.py : _iter_chunks (:213), _hann (:210), _wola_stitch (:227),
      _resample_hq (:159, scipy branch), _from_audio_dict (:125), _make_audio (:116),
      EgregoraAudioSuperResolution surface (:372-386)
  egregora_audio_eval_pack.py        : _stft_mag (:389), _lsd (:405), _si_sdr (:414)
  egregora_fat_llama_gpu.py / _cpu.py: _to_cs, _normalize_audio_input, _fat_llama_upscale
      (with stub `soundfile` and a recording fake of `fat_llama.audio_fattener.feed`)
"""
import hashlib
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _load(name, fname):
    spec = importlib.util.spec_from_file_location(name, REF / fname)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ---- stub soundfile: the reference hard-imports it; we only need write/read to record ----
class _SFStub(types.ModuleType):
    def __init__(self):
        super().__init__("soundfile")
        self.writes = []
        self.next_read = None

    def write(self, path, data, sr, *a, **k):
        self.writes.append((str(path), np.array(data, copy=True), int(sr), a, dict(k)))

    def read(self, path, dtype="float32", always_2d=False):
        y, sr = self.next_read
        return np.asarray(y, dtype=dtype), sr


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    sfstub = _SFStub()
    sys.modules["soundfile"] = sfstub
    sr_mod = _load("ref_sr", "egregora_audio_super_resolution.py")
    ev_mod = _load("ref_eval", "egregora_audio_eval_pack.py")
    gpu_mod = _load("ref_flgpu", "egregora_fat_llama_gpu.py")
    cpu_mod = _load("ref_flcpu", "egregora_fat_llama_cpu.py")

    WIN, HOP = 245760, 221760
    meta = {"numpy": np.__version__, "torch": torch.__version__}
    try:
        import scipy
        meta["scipy"] = scipy.__version__
    except Exception:
        pass

    # ---------------- G1: chunk spans ----------------
    totals = [1, 1000, 245759, 245760, 245761, 467520, 467521, 576000, 2880000, 28800000, 86400000]
    g1 = {}
    for t in totals:
        spans = sr_mod._iter_chunks(t, WIN, HOP)
        if len(spans) <= 16:
            g1[str(t)] = {"n": len(spans), "spans": [list(map(int, s)) for s in spans]}
        else:
            g1[str(t)] = {"n": len(spans), "first": list(map(int, spans[0])),
                          "second": list(map(int, spans[1])), "last": list(map(int, spans[-1])),
                          "sum_start": int(sum(s for s, _ in spans)), "sum_len": int(sum(l for _, l in spans))}
    # odd win/hop combos too (the function is generic)
    g1_generic = []
    for (t, w, h) in [(10, 4, 2), (10, 4, 3), (9, 4, 4), (3, 4, 2), (0, 4, 2), (8, 4, 2), (7, 3, 1)]:
        g1_generic.append({"total": t, "win": w, "hop": h,
                           "spans": [list(map(int, s)) for s in sr_mod._iter_chunks(t, w, h)]})

    # ---------------- G2: hann ----------------
    hann = sr_mod._hann(WIN)
    g2 = {"first8": hann[:8].tolist(), "last8": hann[-8:].tolist(), "mid": float(hann[WIN // 2]),
          "sum_f64": float(hann.astype(np.float64).sum()), "sha256": sha(hann), "dtype": str(hann.dtype),
          "small": {str(n): sr_mod._hann(n).tolist() for n in (1, 2, 3, 8, 9)}}

    # ---------------- G3: WOLA ----------------
    rng = np.random.Generator(np.random.PCG64(1))
    total = 576000
    x = rng.standard_normal((2, total)).astype(np.float32)
    spans = sr_mod._iter_chunks(total, WIN, HOP)
    preds = []
    for s, L in spans:
        c = x[:, s:s + L]
        if L < WIN:
            c = np.concatenate([c, np.zeros((2, WIN - L), np.float32)], axis=1)
        preds.append((c, s, L))
    ident = sr_mod._wola_stitch(preds, total, WIN)
    # random per-chunk predictions with L_pred != L_in (longer and shorter)
    rng2 = np.random.Generator(np.random.PCG64(2))
    preds2 = []
    lp = [WIN + 960, WIN, WIN - 5000]
    for i, (s, L) in enumerate(spans):
        y = rng2.standard_normal((2, lp[i % 3])).astype(np.float32)
        preds2.append((y, s, L))
    rnd = sr_mod._wola_stitch(preds2, total, WIN)
    dec = 4801
    np.savez_compressed(OUT / "g3_wola.npz",
                        total=total, win=WIN, hop=HOP, dec=dec,
                        ident_dec=ident[:, ::dec], ident_head=ident[:, :16], ident_tail=ident[:, -16:],
                        rnd_dec=rnd[:, ::dec], rnd_head=rnd[:, :16], rnd_tail=rnd[:, -16:],
                        lp=np.array(lp))
    g3 = {"ident_sha256": sha(ident), "rnd_sha256": sha(rnd), "seed_x": 1, "seed_pred": 2,
          "ident_max_err_excl_edges": float(np.max(np.abs(ident[:, 1:-1] - x[:, 1:-1]))),
          "ident_first": ident[:, 0].tolist(), "ident_last": ident[:, -1].tolist(),
          "empty": sr_mod._wola_stitch([], 5, WIN).tolist()}
    # small-case WOLA (full output stored) with win=8,hop=6 for kernel unit tests
    rng3 = np.random.Generator(np.random.PCG64(3))
    small = {}
    for (t, w, h) in [(20, 8, 6), (8, 8, 6), (5, 8, 6), (21, 8, 5), (64, 16, 12)]:
        sp = sr_mod._iter_chunks(t, w, h)
        pr = [(rng3.standard_normal((3, w)).astype(np.float32), s, L) for s, L in sp]
        o = sr_mod._wola_stitch(pr, t, w)
        small[f"{t}_{w}_{h}"] = {"spans": [list(map(int, s)) for s in sp],
                                  "preds": [p[0].tolist() for p in pr], "out": o.tolist()}
    g3["small"] = small

    # ---------------- G4: resample (scipy polyphase branch) ----------------
    rng4 = np.random.Generator(np.random.PCG64(4))
    g4 = {}
    arrs = {}
    for (src, dst) in [(44100, 48000), (48000, 96000), (48000, 44100), (16000, 48000)]:
        n = src // 4  # 0.25 s
        noise = (0.25 * rng4.standard_normal((2, n))).astype(np.float32)
        tt = np.arange(n) / src
        sine = (0.5 * np.sin(2 * np.pi * 1000.0 * tt)).astype(np.float32)[None, :]
        yn = sr_mod._resample_hq(noise, src, dst)
        ys = sr_mod._resample_hq(sine, src, dst)
        key = f"{src}_{dst}"
        g4[key] = {"noise_shape": list(yn.shape), "sine_shape": list(ys.shape),
                   "noise_sha256": sha(yn), "sine_sha256": sha(ys), "dtype": str(yn.dtype)}
        arrs[f"in_noise_{key}"] = noise
        arrs[f"out_noise_{key}"] = yn
        arrs[f"out_sine_{key}"] = ys[:, ::37]
    same = sr_mod._resample_hq(np.ones((1, 10), np.float64), 48000, 48000)
    g4["same_sr"] = {"dtype": str(same.dtype), "shape": list(same.shape)}
    np.savez_compressed(OUT / "g4_resample.npz", **arrs)

    # ---------------- G5: shape heuristics ----------------
    g5 = {"from_audio_dict": [], "to_cs": [], "errors": []}
    for shp in [(5,), (9, 8), (8, 8), (8, 9), (2, 100), (100, 2), (100, 9), (3, 2), (2, 3, 4), (1, 1)]:
        a = np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) / 1000.0
        cs, sr = sr_mod._from_audio_dict((a, 44100.0))
        g5["from_audio_dict"].append({"in_shape": list(shp), "out_shape": list(cs.shape),
                                      "sr": sr, "sha256": sha(cs), "dtype": str(cs.dtype)})
    for shp in [(5,), (9, 8), (8, 8), (8, 9), (2, 100), (100, 2), (100, 9), (3, 2), (2, 3, 4), (0,)]:
        for scale in (0.001, 3.0):
            a = (np.arange(int(np.prod(shp)), dtype=np.float32).reshape(shp) - 3.0) * scale
            for nm, m in (("gpu", gpu_mod), ("cpu", cpu_mod)):
                cs = m._to_cs(a)
                g5["to_cs"].append({"mod": nm, "in_shape": list(shp), "scale": scale,
                                    "out_shape": list(cs.shape), "sha256": sha(cs),
                                    "max": float(np.max(np.abs(cs))) if cs.size else 0.0})
    # dict forms
    for shp in [(1, 2, 50), (2, 50), (3, 2, 50)]:
        wf = torch.arange(int(np.prod(shp)), dtype=torch.float64).reshape(shp) / 100.0
        cs, sr = sr_mod._from_audio_dict({"waveform": wf, "sample_rate": 48000.0})
        g5["from_audio_dict"].append({"dict_shape": list(shp), "out_shape": list(cs.shape), "sr": sr,
                                      "sha256": sha(cs), "dtype": str(cs.dtype)})
    for bad in [torch.zeros(5), torch.zeros(1, 1, 2, 5)]:
        for nm, fn in (("sr", lambda w: sr_mod._from_audio_dict({"waveform": w, "sample_rate": 1})),
                       ("gpu", lambda w: gpu_mod._normalize_audio_input({"waveform": w, "sample_rate": 1})),
                       ("cpu", lambda w: cpu_mod._normalize_audio_input({"waveform": w, "sample_rate": 1}))):
            try:
                fn(bad)
                g5["errors"].append({"mod": nm, "shape": list(bad.shape), "raised": None})
            except Exception as e:
                g5["errors"].append({"mod": nm, "shape": list(bad.shape), "raised": type(e).__name__,
                                     "msg": str(e)})
    for nm, fn in (("sr", lambda: sr_mod._from_audio_dict(None)),
                   ("gpu", lambda: gpu_mod._normalize_audio_input(None, "", "")),
                   ("cpu", lambda: cpu_mod._normalize_audio_input(None, "", "")),
                   ("gpu_path", lambda: gpu_mod._normalize_audio_input(None, "/nonexistent/x.wav", ""))):
        try:
            fn()
        except Exception as e:
            g5["errors"].append({"mod": nm, "case": "none", "raised": type(e).__name__, "msg": str(e)})

    # ---------------- G6: _make_audio ----------------
    a = np.arange(12, dtype=np.float64).reshape(2, 6)
    d = sr_mod._make_audio(44100.0, a)
    d1 = sr_mod._make_audio(8000, np.arange(4))
    g6 = {"shape": list(d["waveform"].shape), "dtype": str(d["waveform"].dtype),
          "contig": bool(d["waveform"].is_contiguous()), "sr": d["sample_rate"],
          "sr_type": type(d["sample_rate"]).__name__, "shape_1d": list(d1["waveform"].shape),
          "keys": sorted(d.keys())}

    # ---------------- G7: STFT magnitude / LSD / SI-SDR ----------------
    rng7 = np.random.Generator(np.random.PCG64(7))
    n7 = 24000
    t7 = np.arange(n7) / 48000.0
    sig = (0.3 * np.sin(2 * np.pi * 440 * t7) + 0.1 * np.sin(2 * np.pi * 5000 * t7)
           + 0.05 * rng7.standard_normal(n7)).astype(np.float32)
    sig2 = np.stack([sig, np.roll(sig, 7) * 0.5]).astype(np.float32)
    S = ev_mod._stft_mag(sig)
    S2 = ev_mod._stft_mag(sig2)
    Sshort = ev_mod._stft_mag(sig[:1000])
    S_1024_256 = ev_mod._stft_mag(sig[:20000], n_fft=1024, hop=256)
    pert = (sig + 1e-3 * rng7.standard_normal(n7)).astype(np.float32)
    Sp = ev_mod._stft_mag(pert)
    lsd_self = ev_mod._lsd(S, S)
    lsd_gain = ev_mod._lsd(S, ev_mod._stft_mag((sig * np.float32(1.001)).astype(np.float32)))
    lsd_pert = ev_mod._lsd(S, Sp)
    np.savez_compressed(OUT / "g7_stft.npz", sig=sig, sig2=sig2, pert=pert, S=S, S2=S2, Sshort=Sshort,
                        S_1024_256=S_1024_256, Sp=Sp)
    g7 = {"S_shape": list(S.shape), "S_dtype": str(S.dtype), "lsd_self": list(lsd_self),
          "lsd_gain_1p001": list(lsd_gain), "lsd_pert": list(lsd_pert),
          "si_sdr_self": float(ev_mod._si_sdr(sig, sig)), "si_sdr_pert": float(ev_mod._si_sdr(sig, pert)),
          "si_sdr_stereo": float(ev_mod._si_sdr(sig2, sig2[:, :20000] * 0.9)),
          "Sshort_shape": list(Sshort.shape)}

    # ---------------- G8: node surface ----------------
    def surf(cls):
        import inspect
        it = cls.INPUT_TYPES()
        return {"INPUT_TYPES": it, "widget_order": {k: list(v.keys()) for k, v in it.items()}, "RETURN_TYPES": list(cls.RETURN_TYPES),
                "FUNCTION": cls.FUNCTION, "CATEGORY": cls.CATEGORY, "OUTPUT_NODE": cls.OUTPUT_NODE,
                "run_signature": str(inspect.signature(cls.run))}
    g8 = {
        "EgregoraAudioUpscaler": surf(sr_mod.EgregoraAudioSuperResolution),
        "EgregoraFatLlamaGPU": surf(gpu_mod.EgregoraFatLlamaGPU),
        "EgregoraFatLlamaCPU": surf(cpu_mod.EgregoraFatLlamaCPU),
        "display": {**sr_mod.NODE_DISPLAY_NAME_MAPPINGS, **gpu_mod.NODE_DISPLAY_NAME_MAPPINGS,
                    **cpu_mod.NODE_DISPLAY_NAME_MAPPINGS},
        "class_names": {"EgregoraAudioUpscaler": "EgregoraAudioSuperResolution",
                        "EgregoraFatLlamaGPU": "EgregoraFatLlamaGPU",
                        "EgregoraFatLlamaCPU": "EgregoraFatLlamaCPU"},
        "runner_consts": {"REQ_SR": sr_mod._FlashSRRunner.REQ_SR, "CHUNK_S": sr_mod._FlashSRRunner.CHUNK_S,
                          "OVERLAP_S": sr_mod._FlashSRRunner.OVERLAP_S,
                          "CHUNK_SAMPLES": sr_mod._FlashSRRunner.CHUNK_SAMPLES,
                          "HOP": int((sr_mod._FlashSRRunner.CHUNK_S - sr_mod._FlashSRRunner.OVERLAP_S) * 48000)},
    }

    # ---------------- G9: Fat-Llama adapter contract via a recording fake feed ----------------
    g9 = {}
    for nm, m, pkg in (("gpu", gpu_mod, "fat_llama"), ("cpu", cpu_mod, "fat_llama_fftw")):
        feed = types.ModuleType(f"{pkg}.audio_fattener.feed")
        rec = {"upscale": None, "writes": []}

        class _Audio:
            sample_width = 2

        def read_audio(file_path, format, _rec=rec):
            return 48000, np.zeros(4, np.int16), 1536000, _Audio()

        def write_audio(file_path, sample_rate, data, format, _rec=rec):
            _rec["writes"].append(np.array(data, copy=True))

        def upscale(**kw):
            rec["upscale"] = dict(kw)

        feed.read_audio, feed.write_audio, feed.upscale = read_audio, write_audio, upscale
        p0 = types.ModuleType(pkg)
        p1 = types.ModuleType(f"{pkg}.audio_fattener")
        p1.feed = feed
        p0.audio_fattener = p1
        sys.modules[pkg], sys.modules[f"{pkg}.audio_fattener"] = p0, p1
        sys.modules[f"{pkg}.audio_fattener.feed"] = feed
        if nm == "gpu":
            m._fat_llama_upscale(Path("/tmp/in.wav"), Path("/tmp/out.flac"), "flac", 77, 0.25, 1536, True, False)
        else:
            m._fat_llama_fftw_upscale(Path("/tmp/in.wav"), Path("/tmp/out.flac"), "flac", 77, 0.25, 1536)
        entry = {"upscale_kwargs": rec["upscale"]}
        # exercise patched write_audio
        cases = {}
        for cname, data, do_read in (("le1", np.array([0.5, -1.0, 0.25], np.float32), True),
                                     ("gt1_sw2", np.array([16384.0, -32768.0, 100.0], np.float32), True),
                                     ("gt1_swNone", np.array([16384.0, -32768.0, 100.0], np.float32), False)):
            rec["writes"].clear()
            if do_read:
                feed.read_audio("/tmp/in.wav", "wav")
            else:
                feed._egregora_sample_width = None
            feed.write_audio("/tmp/o.wav", 48000, data, "wav")
            cases[cname] = rec["writes"][-1].astype(np.float64).tolist()
        entry["write_patch"] = cases
        g9[nm] = entry
    # dict path writes cs.T to soundfile with sr; record what was handed over
    sfstub.writes.clear()
    wf = torch.tensor([[[0.1, -0.2, 0.3], [1.5, -2.5, 0.0]]], dtype=torch.float32)
    cs, sr, p = gpu_mod._normalize_audio_input({"waveform": wf, "sample_rate": 44100})
    w = sfstub.writes[-1]
    g9["dict_path"] = {"cs_shape": list(cs.shape), "sr": sr, "written_shape": list(w[1].shape),
                       "written": w[1].astype(np.float64).tolist(), "written_sr": w[2],
                       "write_extra_args": [list(w[3]), w[4]], "note": "no clamp/normalise on dict path"}
    sfstub.writes.clear()
    cs, sr, p = gpu_mod._normalize_audio_input((np.array([[0.5, 2.0], [-4.0, 1.0], [0.25, 0.0]]), 22050))
    w = sfstub.writes[-1]
    g9["tuple_path"] = {"cs_shape": list(cs.shape), "sr": sr, "written": w[1].astype(np.float64).tolist()}

    for name, obj in (("g1_chunks", {"win": WIN, "hop": HOP, "totals": g1, "generic": g1_generic}),
                      ("g2_hann", g2), ("g3_wola", g3), ("g4_resample", g4), ("g5_shapes", g5),
                      ("g6_make_audio", g6), ("g7_metrics", g7), ("g8_surface", g8), ("g9_fatllama_adapter", g9),
                      ("meta", meta)):
        (OUT / f"{name}.json").write_text(json.dumps(obj, indent=1, sort_keys=True, ensure_ascii=False) + "\n",
                                          encoding="utf-8")
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
