#!/usr/bin/env python3
"""Opt-in external check (never run by the tests or the bench): compare this pack's Fat-Llama node arithmetic with the
UPSTREAM packages the reference delegates to, on a machine that has them installed.

  pip install fat-llama-fftw soundfile pydub        # (and ffmpeg on PATH), on a box with an MI355X + this pack built
  python tools/compare_with_upstream.py [--seconds 10] [--sr 16000] [--iters 50] [--kbps 1411]

It writes a seeded WAV, runs `fat_llama_fftw.audio_fattener.feed.upscale(...)` with exactly the 7 kwargs the
reference's CPU node passes (reference egregora_fat_llama_cpu.py:126-134), runs the device engine through the node
(`EgregoraFatLlamaGPU`), and prints the reference's own LSD / SI-SDR metric between the two results plus the
fraction of PCM_16 samples that differ.  Until someone runs this, parity with upstream is UNPINNED (see oracle/fatllama.py, SPEC.md):
every disagreement maps to one named field of oracle.fatllama.FatLlamaSpec (factor rounding, interpolation kernel, threshold
reference / kind, autoscale / normalise definitions, PCM scales); `--variants` sweeps the device-side readings of SPEC.md
section 3.  `--flashsr CKPT_DIR` does the same for FlashSR through the node's own checkpoint loader (flashsr_weights.load).
"""
import argparse
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--thr", type=float, default=0.6)
    ap.add_argument("--kbps", type=int, default=1411)
    ap.add_argument("--variants", action="store_true", help="sweep the threshold / interpolation readings of SPEC.md section 3")
    ap.add_argument("--flashsr", default="", metavar="CKPT_DIR",
                    help="instead: compare FlashSR_Inference (importable) with this pack on one seeded chunk; CKPT_DIR holds "
                         "student_ldm.pth / sr_vocoder.pth / vae.pth and is read through flashsr_weights.load (the node's loader)")
    args = ap.parse_args()
    if args.flashsr:
        return compare_flashsr(args.flashsr)
    try:
        import soundfile as sf
        from fat_llama_fftw.audio_fattener import feed
    except Exception as e:
        sys.exit(f"upstream packages not importable here ({e}); this tool is opt-in")
    import torch
    from packload import load_pack
    from oracle import metrics as om
    pack = load_pack()

    n = int(args.seconds * args.sr)
    rng = np.random.Generator(np.random.PCG64(101))
    t = np.arange(n) / args.sr
    x = sum(np.sin(2 * np.pi * f * t) / (k + 1) for k, f in enumerate(np.geomspace(80, 6000, 8))) + 0.01 * rng.standard_normal(n)
    x = (0.5 * x / np.max(np.abs(x))).astype(np.float32)
    tmp = Path(tempfile.mkdtemp())
    sf.write(str(tmp / "in.wav"), x, args.sr)
    feed.upscale(input_file_path=str(tmp / "in.wav"), output_file_path=str(tmp / "up.wav"), source_format="wav",
                 target_format="wav", max_iterations=args.iters, threshold_value=args.thr, target_bitrate_kbps=args.kbps)
    up, sr_up = sf.read(str(tmp / "up.wav"), dtype="float32", always_2d=False)

    import itertools
    import os
    node = pack.NODE_CLASS_MAPPINGS["EgregoraFatLlamaCPU"]()          # same 7-kwarg contract, device engine
    combos = [""]
    if args.variants:
        names = ("relative", "soft", "no_init_thr")
        thr = [",".join(c) for r in range(len(names) + 1) for c in itertools.combinations(names, r)]
        # x every up-rating rule: linear by the integer factor (survey recall), zero insertion, numpy.interp on the endpoint-inclusive
        # linspace grid by the integer factor, and the same with the ratio applied before int() (judge recall, SPEC.md section 3)
        combos = [",".join(t for t in (a, b) if t) for a in thr for b in ("", "zero_stuff", "linspace", "linspace,ratio_then_int")]
    for combo in combos:
        os.environ["EGREGORA_FATLLAMA_SPEC"] = combo
        (res,) = node.run("wav", args.iters, args.thr, args.kbps, AUDIO={"waveform": torch.from_numpy(x)[None, None], "sample_rate": args.sr})
        mine = res["waveform"][0, 0].numpy()
        m = min(len(up), len(mine))
        lsd = om.lsd_audio(up[:m], mine[:m])
        print(f"[{combo or 'default'}] upstream {up.shape} @ {sr_up} Hz vs this pack {mine.shape} @ {res['sample_rate']} Hz: "
              f"LSD mean/p95 = {lsd[0]:.4g} / {lsd[1]:.4g} dB   SI-SDR = {om.si_sdr(up[:m], mine[:m]):.2f} dB   "
              f"PCM_16 samples differing = {np.mean(np.abs(up[:m] - mine[:m]) * 32768 > 0.5):.4f}")


def compare_flashsr(ckpt_dir):
    """Upstream FlashSR vs this pack on one 5.12 s chunk with the SAME injected noise (upstream draws its own, so its sampler is
    patched to return ours); prints per-stage shapes that differ and the LSD of the waveforms."""
    import torch
    try:
        from FlashSR.FlashSR import FlashSR
    except Exception as e:
        sys.exit(f"FlashSR_Inference not importable here ({e}); this tool is opt-in")
    from packload import load_pack
    load_pack()
    from egregora_amd import flashsr_engine as E, flashsr_weights as W
    from oracle import metrics as om
    d = Path(ckpt_dir)
    for f in W.FILES:
        print(f"# {f}: {len(W.read_state_dict(d / f))} tensors")
    params, cfg, _ = W.load(d)                      # raises with the list of unmapped / mismatched tensors: fix flashsr_keymap.json
    eng = E.FlashSREngine(cfg, params)
    g = torch.Generator().manual_seed(7)
    x = (0.3 * torch.randn(1, cfg.chunk, generator=g)).cuda()
    nz = eng.noise(1, None, 7)
    mine = eng.c_forward(x, nz).cpu().numpy()
    model = FlashSR(str(d / "student_ldm.pth"), str(d / "sr_vocoder.pth"), str(d / "vae.pth")).eval().cuda()
    real_randn = torch.randn
    torch.randn = lambda *a, **k: nz.permute(0, 3, 1, 2).contiguous() if tuple(a[:1]) and list(a[0] if isinstance(a[0], (tuple, list)) else a) == list(nz.permute(0, 3, 1, 2).shape) else real_randn(*a, **k)
    try:
        with torch.inference_mode():
            up = model(x, lowpass_input=False).float().cpu().numpy()
    finally:
        torch.randn = real_randn
    m = min(up.shape[-1], mine.shape[-1])
    lsd = om.lsd_audio(up[..., :m], mine[..., :m])
    print(f"FlashSR upstream vs this pack: LSD mean/p95 = {lsd[0]:.4g} / {lsd[1]:.4g} dB, SI-SDR {om.si_sdr(up[0, :m], mine[0, :m]):.2f} dB")


if __name__ == "__main__":
    main()
