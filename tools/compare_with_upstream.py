#!/usr/bin/env python3
"""Opt-in external check (never run by the tests or the bench): compare this pack's Fat-Llama node arithmetic with the
UPSTREAM packages the reference delegates to, on a machine that has them installed.

  pip install fat-llama-fftw soundfile pydub        # (and ffmpeg on PATH), on a box with an MI355X + this pack built
  python tools/compare_with_upstream.py [--seconds 10] [--sr 16000] [--iters 50] [--kbps 1411]

It writes a seeded WAV, runs `fat_llama_fftw.audio_fattener.feed.upscale(...)` with exactly the 7 kwargs the
reference's CPU node passes (reference egregora_fat_llama_cpu.py:126-134), runs the device engine through the node
(`EgregoraFatLlamaGPU`), and prints the reference's own LSD / SI-SDR metric between the two results plus the
fraction of PCM_16 samples that differ.  Until someone runs this, parity with upstream is UNPINNED (see oracle/fatllama.py):
every disagreement maps to one named field of oracle.fatllama.FatLlamaSpec (factor rounding, interpolation kernel,
autoscale / normalise definitions, PCM scales).
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
    args = ap.parse_args()
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

    node = pack.NODE_CLASS_MAPPINGS["EgregoraFatLlamaCPU"]()          # same 7-kwarg contract, device engine
    (res,) = node.run("wav", args.iters, args.thr, args.kbps, AUDIO={"waveform": torch.from_numpy(x)[None, None], "sample_rate": args.sr})
    mine = res["waveform"][0, 0].numpy()
    print(f"upstream: {up.shape} @ {sr_up} Hz   this pack: {mine.shape} @ {res['sample_rate']} Hz")
    m = min(len(up), len(mine))
    lsd = om.lsd_audio(up[:m], mine[:m])
    print(f"LSD mean/p95 = {lsd[0]:.4g} / {lsd[1]:.4g} dB   SI-SDR = {om.si_sdr(up[:m], mine[:m]):.2f} dB   "
          f"PCM_16 samples differing = {np.mean(np.abs(up[:m] - mine[:m]) * 32768 > 0.5):.4f}")


if __name__ == "__main__":
    main()
