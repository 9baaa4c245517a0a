"""Command-line front end with the flag set of the reference's `flashsr_min.py` (:6-12):

    python flashsr_min.py --ckpt-dir DIR --in in.wav --out out.wav [--target-sr 48000] [--device auto]

The reference file is an inert stub (it copies the input).  This one runs the real path: WAV in ->
EgregoraAudioSuperResolution node (resample to 48 kHz, chunked FlashSR on the MI355X, WOLA, resample to --target-sr)
-> PCM_16 WAV out, then prints OK like the reference.  `--ckpt-dir` is where the three upstream checkpoints are looked for first
(student_ldm.pth, sr_vocoder.pth, vae.pth -- flashsr_weights.py); it may instead hold `flashsr_amd_state_dict.pt` (weights
already in flashsr_arch names).
"""
import argparse
import importlib.util
import os
import sys
from pathlib import Path


def _load_pack():
    here = Path(__file__).resolve().parent
    spec = importlib.util.spec_from_file_location("egregora_amd", here / "__init__.py", submodule_search_locations=[str(here)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["egregora_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt-dir", required=True)
    ap.add_argument("--in", dest="inp", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--target-sr", type=int, default=48000)
    ap.add_argument("--device", default="auto")
    ap.add_argument("--lowpass-input", action="store_true")
    args = ap.parse_args(argv)
    if args.device not in ("auto", "cuda"):
        raise SystemExit("this pack has no CPU path: --device must be auto or cuda (an MI355X)")
    if args.target_sr not in (48000, 44100, 96000):
        raise SystemExit("--target-sr must be 48000, 44100 or 96000 (the node's output_sr choices)")
    ck = Path(args.ckpt_dir) / "flashsr_amd_state_dict.pt"
    if ck.exists():
        os.environ.setdefault("EGREGORA_FLASHSR_WEIGHTS", str(ck))
    else:
        os.environ.setdefault("EGREGORA_FLASHSR_CKPT_DIR", str(args.ckpt_dir))
    pack = sys.modules.get("egregora_amd") or _load_pack()
    from egregora_amd import wavio
    wav, sr = wavio.read_wav(args.inp)                       # [S] or [S,C], like sf.read(always_2d=False)
    node = pack.NODE_CLASS_MAPPINGS["EgregoraAudioUpscaler"]()
    (res,) = node.run(audio=(wav, sr), lowpass_input=bool(args.lowpass_input), output_sr=str(args.target_sr))
    out = res["waveform"][0].numpy().T                       # [S,C]
    wavio.write_wav_pcm16(args.out, out, res["sample_rate"])
    print("OK")


if __name__ == "__main__":
    main()
