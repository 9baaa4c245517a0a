"""BASELINE.json configs[3] and configs[4] at their full sizes on ONE GPU, through the ComfyUI node surface, checked by
size-independent properties (the oracle cannot run these sizes in seconds):

  C4  FlashSR long-form, 10 min stereo @48 kHz: 130 chunks (260 batch rows, 9 engine passes), WOLA; output shape / rate,
      finiteness, reference quirk Q1 (first sample exactly 0), run-to-run determinism, and independence of the result (to fp32
      round-off) from how the chunk list is split into engine passes -- the property the 8-GPU sharding relies on.
  C5  the full chain 30 min 44.1 kHz -> [DeepFilterNet stage bypassed: upstream model absent] -> FlashSR (output_sr 96000)
      -> Fat-Llama 200 iterations, target_bitrate_kbps 3072 (factor 1): rates and lengths through both rate conversions, the
      three-level Fat-Llama plan at N = 172.8 M samples per channel, PCM_16 grid and peak bound of the node output.
Synthetic weights (declared architecture) as everywhere in this build; timings are printed for DESIGN.md.
"""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def synth(seed, n, sr, channels=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64) / sr
    x = np.stack([sum(np.sin(2 * np.pi * f * (1 + 0.013 * c) * t + rng.uniform(0, 6.28)) / (k + 1)
                      for k, f in enumerate(np.geomspace(80.0, 6000.0, 8))) + 0.01 * rng.standard_normal(n)
                  for c in range(channels)])
    return (0.5 * x / np.max(np.abs(x))).astype(np.float32)


@pytest.fixture(scope="module")
def full_engine(pack):
    from egregora_amd import flashsr_arch as A, flashsr_engine as E
    cfg = A.FlashSRConfig()
    eng = E.FlashSREngine(cfg, A.init_params(cfg, 0))
    E.set_engine(eng)
    yield eng
    E.set_engine(None)


def test_c4_ten_minutes_stereo_long_form(pack, full_engine):
    from egregora_amd import audio_glue as ag, flashsr_engine as E
    total = 600 * 48000
    x = synth(404, total, 48000)
    assert len(ag.spans(total)) == 130
    node = pack.NODE_CLASS_MAPPINGS["EgregoraAudioUpscaler"]()
    A = {"waveform": torch.from_numpy(x)[None], "sample_rate": 48000}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (out,) = node.run(A, False, "48000")
    dt = time.perf_counter() - t0
    y = out["waveform"]
    print(f"\nC4 on one MI355X: 600 s stereo in {dt:.2f} s = {600 / dt:.0f} xRT (AUDIO dict in -> AUDIO dict out, PCIe included)")
    assert tuple(y.shape) == (1, 2, total) and out["sample_rate"] == 48000 and y.dtype == torch.float32
    assert bool(torch.isfinite(y).all()) and float(y.abs().max()) > 0
    assert float(y[0, :, 0].abs().max()) == 0.0                       # Q1: Hann endpoint
    # every call runs the two-term fp16 kernels with per-row device-side scales (include/egregora_amd.h egr_flashsr_set_split):
    # the result does not depend on the handle's history, so calls 1, 2 and 3 are bit-identical; the first also pays the scratch
    # arena's allocations
    (out2,) = node.run(A, False, "48000")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (out2b,) = node.run(A, False, "48000")
    dt2 = time.perf_counter() - t0
    print(f"C4 on one MI355X, steady state (fp16 operand terms): 600 s stereo in {dt2:.2f} s = {600 / dt2:.0f} xRT")
    assert torch.equal(out2b["waveform"], out2["waveform"])           # deterministic
    d12 = float((out2["waveform"] - y).abs().max())
    print(f"C4: max |difference| between the first call and the calls after it {d12:.2e} (peak {float(y.abs().max()):.3f}); {full_engine.split_info()}")
    assert d12 == 0.0
    old = E.ROWS_PER_PASS
    try:
        E.ROWS_PER_PASS = 14                                          # 19 passes instead of 9, different pass boundaries
        (out3,) = node.run(A, False, "48000")
    finally:
        E.ROWS_PER_PASS = old
    # rows are independent; only tile / split-K choices (which follow the row count of a pass) move the fp32 round-off
    d = float((out3["waveform"] - y).abs().max())
    print(f"C4: max |difference| between 32-row and 14-row passes {d:.2e} (peak {float(y.abs().max()):.3f})")
    assert d <= 2e-4 * float(y.abs().max())


def test_c5_full_chain_thirty_minutes(pack, full_engine):
    n_in = 30 * 60 * 44100
    x = synth(505, n_in, 44100)
    up = pack.NODE_CLASS_MAPPINGS["EgregoraAudioUpscaler"]()
    fl = pack.NODE_CLASS_MAPPINGS["EgregoraFatLlamaGPU"]()
    A = {"waveform": torch.from_numpy(x)[None], "sample_rate": 44100}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (mid,) = up.run(A, False, "96000")
    t1 = time.perf_counter()
    assert mid["sample_rate"] == 96000 and tuple(mid["waveform"].shape) == (1, 2, 172800000)      # 44.1k -> 48k -> 96k
    (out,) = fl.run("wav", 200, 0.6, 3072, True, False, AUDIO=mid)
    t2 = time.perf_counter()
    print(f"\nC5 on one MI355X (DeepFilterNet stage bypassed): FlashSR {t1 - t0:.2f} s + Fat-Llama(200) {t2 - t1:.2f} s "
          f"for 1800 s of audio = {1800 / (t2 - t0):.0f} xRT")
    y = out["waveform"]
    assert out["sample_rate"] == 96000 and tuple(y.shape) == (1, 2, 172800000) and y.dtype == torch.float32
    assert bool(torch.isfinite(y).all())
    peak = float(y.abs().max())
    assert 0.0 < peak <= 1.0                                           # normalised, then PCM_16
    q = y[0, :, ::4801] * 32768.0
    assert float((q - q.round()).abs().max()) == 0.0                   # values sit on the PCM_16 grid (k / 32768)
