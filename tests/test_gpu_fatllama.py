"""GPU parity: HIP Fat-Llama engine (through the C ABI) vs the oracle restatement, same seeded inputs.

Tolerances (floating point; stated per north_star "within 1e-3 LSD"):
  * raw loop output: max|gpu-oracle| <= 2e-5 * max|oracle| (float32 FFT round-off, both sides float32), and
    the GPU's error against the float64 run of the same loop is <= 3x the float32 oracle's own error
  * LSD(gpu, oracle) <= 1e-3 dB with the reference's own metric (oracle.metrics.lsd_audio) on full-band
    material.  Where the spectrum has nulls > 100 dB below the peak (linear up-rating puts sinc^2 zeros at
    multiples of the source rate) float32 round-off of EITHER implementation decides those bins: there the
    gate is the LSD of the device against the FLOAT64 run of the same loop over the bins a float32 transform can
    resolve to 1e-3 dB at all -- magnitude >= 80 dB above the float32 ORACLE's measured round-off floor
    (oracle.metrics.lsd_masked; the kept fraction is printed and bounded) -- <= 1e-3 dB, plus
    rms(gpu - f64) <= 2.5 * rms(oracle_f32 - f64) over everything.
  * after the PCM_16 hop (node output): values are k/32768 and the quantiser turns a 3e-7 relative error
    into a +-1 LSB flip wherever x*32767 lands within ~0.01 of a rounding boundary (~2% of samples for ANY
    two float32 pipelines): require |diff| <= 1 LSB everywhere and <= 5% of samples differing.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fatllama as ofl
from oracle import metrics as om

pytestmark = pytest.mark.gpu

# Where the factor > 1 LSD gate applies: bins at least this far above the float32 round-off floor (measured from the float32
# oracle vs float64).  At the margin a float32 result sits 1e-4 relative = 8.7e-4 dB from float64 -- the north star's bar.
F32_MARGIN_DB = 80.0
UNMASKED_LSD_BOUND_DB = 0.3    # unmasked LSD(device, oracle32) for factor > 1: measured 0.03 - 0.10 dB (profiles/r03/pytest_gpu_tail.txt); 3x the largest


def synth(C, n, seed, scale=8000.0, integer=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 48000.0
    x = np.stack([sum(np.sin(2 * np.pi * f * (1 + 0.01 * c) * t + c) / k for k, f in enumerate((110, 440, 1234, 5000, 9000), 1))
                  for c in range(C)])
    x = x / np.max(np.abs(x)) * scale + rng.standard_normal((C, n)) * scale * 0.01
    if integer:
        x = np.rint(x)
        x[:, rng.integers(0, n, n // 50)] = 0.0      # exact zeros: the only samples a 0.6 threshold touches
    return x.astype(np.float32)


def oracle32_at_n_plus_2():
    """(max, rms, plain LSD in dB) of the float32 ORACLE against the float64 loop at N = 2 880 002, 800 iterations: a committed record
    written by tools/probe_c3_plus2_oracle.py (ten minutes of host time: Bluestein in pocketfft), with its seed, box and date."""
    import json
    from pathlib import Path
    r = json.loads((Path(__file__).resolve().parent / "golden" / "oracle32_c3_plus2.json").read_text())
    assert r["n"] == 2880002 and r["iterations"] == 800 and r["seed"] == 2880, r
    return (r["max_err"], r["rms_err"], r["lsd_plain_db"])


def run_gpu(pack, x, factor, iters, thr, normalize=False, autoscale=False, pcm_in=False, node_post=False, **kw):
    from egregora_amd import fatllama_engine as fe
    xt = torch.from_numpy(x).cuda()
    y = fe.enhance_device(xt, factor, iters, thr, normalize, autoscale, pcm_in, node_post, **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy()


CASES = [
    # (C, n_in, factor, iters, thr)
    (1, 8, 1, 1, 0.6),            # M = 4
    (1, 100, 1, 3, 0.6),
    (2, 1200, 1, 5, 0.6),
    (1, 1000, 6, 4, 0.6),         # C1-like up-rate
    (2, 2 * 3 * 5 * 7 * 11 * 13, 1, 3, 0.6),   # every supported odd radix
    (1, 16000, 2, 10, 0.6),
    (2, 48000, 1, 20, 0.6),
    (1, 160000, 6, 5, 0.6),       # C1 shape (N' = 960000), fewer iterations
    (2, 9600, 1, 77, 0.6),        # > 50 iterations: three replays of the captured 25-iteration hipGraph + a remainder
]


@pytest.mark.parametrize("C,n,f,iters,thr", CASES)
def test_loop_matches_oracle(pack, C, n, f, iters, thr):
    x = synth(C, n, seed=n + f)
    want = ofl.enhance_channels(x, f, iters, thr, normalize=False, autoscale=False)
    got = run_gpu(pack, x, f, iters, thr)
    assert got.shape == want.shape
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale
    exact = ofl.enhance_channels(x, f, iters, thr, normalize=False, autoscale=False, exact=True)
    err_oracle = float(np.max(np.abs(want - exact)))
    err_gpu = float(np.max(np.abs(got - exact)))
    assert err_gpu <= 3.0 * err_oracle + 1e-7 * scale, (err_gpu, err_oracle)
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    assert rms(got - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale
    if want.shape[1] >= 4096:
        if f == 1:
            assert om.lsd_audio(want, got)[0] <= 1e-3
        else:
            lg, kept = om.lsd_masked(exact, got, f32_run=want, margin_db=F32_MARGIN_DB)
            lo, _ = om.lsd_masked(exact, want, f32_run=want, margin_db=F32_MARGIN_DB)
            print(f"\nfactor {f}: LSD over the {kept:.1%} of bins >= {F32_MARGIN_DB:.0f} dB above the float32 floor: device {lg:.2e} dB, float32 oracle "
                  f"{lo:.2e} dB; within 100 dB of the peak: {om.lsd_masked(exact, got, 100.0)[0]:.2e} / {om.lsd_masked(exact, want, 100.0)[0]:.2e}")
            assert lg <= 1e-3 and kept >= 0.5, (lg, lo, kept)
            # ... and the UNMASKED metric between the two float32 runs, with its own (looser, stated) bound, so that a regression
            # cannot hide behind the mask: the bins the mask drops are float32 round-off of BOTH runs (two independent noise floors
            # 120+ dB under the peak compared in dB), which is what this number measures
            plain = om.lsd_audio(want, got)[0]
            print(f"factor {f}: unmasked LSD(device, oracle32) {plain:.3e} dB (bound {UNMASKED_LSD_BOUND_DB} dB)")
            assert plain <= UNMASKED_LSD_BOUND_DB, plain


@pytest.mark.parametrize("C,n,iters", [(1, 4800, 51), (2, 9600, 77), (3, 4800, 130)])
def test_graph_replay_equals_plain_launches(pack, C, n, iters):
    """Above 50 iterations the loop body runs as replays of a captured 25-iteration hipGraph (one or two pipelines) plus a
    remainder; profiling runs use plain stream launches.  Same kernels, same order per channel: bit-identical outputs."""
    x = synth(C, n, seed=n + iters)
    a = run_gpu(pack, x, 1, iters, 0.6)
    b = run_gpu(pack, x, 1, iters, 0.6, profile=True)
    a2 = run_gpu(pack, x, 1, iters, 0.6)              # cached executable graph
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, a2)


def test_verified_side_stream_and_graph_switch(pack):
    """The second channel pipeline on a side stream handed in by the host (verified to overlap with the caller's), loop replayed
    from the captured graph or launched plainly: all four combinations give the same bits; long stereo runs tune this once."""
    import ctypes as C
    from egregora_amd import fatllama_engine as fe, native, streams
    x = synth(2, 9600, seed=5)
    base = run_gpu(pack, x, 1, 130, 0.6, profile=True)
    plan = fe._plan(9600, 2, 1, 0)
    side = streams.side_streams(1)
    L = native.lib()
    if side:
        native.check(L.egr_fatllama_set_side_stream(C.c_void_p(plan), C.c_void_p(side[0].cuda_stream)), "set_side_stream")
    xd = torch.from_numpy(x).cuda()
    for mode in (0, 1, 0, 1):
        native.check(L.egr_fatllama_set_graph(C.c_void_p(plan), mode), "set_graph")
        out = torch.empty_like(xd)
        native.check(L.egr_fatllama_enhance(C.c_void_p(plan), native.ptr(xd), native.ptr(out), 130, 0.6, 0, native.stream_ptr()), "enhance")
        np.testing.assert_array_equal(out.cpu().numpy(), base)
    np.testing.assert_array_equal(run_gpu(pack, x, 1, 130, 0.6), base)          # the engine's own call: tunes this plan once
    assert (plan, torch.cuda.current_stream().cuda_stream) in fe._SIDE_SET


@pytest.mark.parametrize("n,split", [(420, (6, 5, 7)), (2 * 8 * 9 * 10, (8, 9, 10)), (2 * 16 * 15 * 64, (16, 15, 64)),
                                     (2 * 4 * 3 * 5, (4, 3, 5)), (48000, (20, 24, 50)), (48000, (40, 600, 1))])
def test_three_level_plan_matches_oracle(pack, n, split):
    """Explicit 3-level factorisations (the path long inputs take) against the oracle at small sizes."""
    x = synth(2, n, seed=n)
    want = ofl.enhance_channels(x, 1, 4, 0.6, normalize=False, autoscale=False)
    got = run_gpu(pack, x, 1, 4, 0.6, split=split)
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale


@pytest.mark.parametrize("n,split,levels", [(9600000, (1875, 2560, 1), 2), (9600000, None, 3), (19200000, None, 3)])
def test_long_inputs_wide_columns_and_three_levels(pack, n, split, levels):
    """200 s mono at 48 kHz as two levels with 1875-point outer columns on 4-column tiles (above the 1024 points of an 8-column
    tile; the planner's choice until round 4, now an explicit split) and as the planner plans it today -- three levels around the
    two-barrier kernels, 625 x 2 x 3840; 400 s (9.6 M complex points > 2048 x 4096) takes three levels either way.  2 iterations,
    vs the oracle."""
    from egregora_amd import fatllama_engine as fe
    info = fe.plan_info(n, 1)
    assert info["supported"] and info["M1"] * info["M2"] * info["M3"] == n // 2
    if split is None:
        assert info["levels"] == levels and info["M1"] == 625, info
    x = synth(1, n, seed=77)
    want = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False)
    got = run_gpu(pack, x, 1, 2, 0.6, **({"split": split} if split else {}))
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale
    assert om.lsd_audio(want[:, :960000], got[:, :960000])[0] <= 1e-3


@pytest.mark.parametrize("thr", [50.0, 3000.0])
def test_large_threshold_actually_gates_bins(pack, thr):
    """With a threshold inside the data range a sizeable share of samples/bins is zeroed; a borderline
    bin may legitimately flip, so compare in the LSD sense and in energy."""
    x = synth(1, 4800, seed=5, scale=100.0)
    want = ofl.enhance_channels(x, 1, 4, thr, normalize=False, autoscale=False)
    got = run_gpu(pack, x, 1, 4, thr)
    num = float(np.sum((got - want) ** 2)); den = float(np.sum(want ** 2)) + 1e-30
    assert num / den < 1e-6


@pytest.mark.parametrize("C,n,sr,kbps", [(1, 16000, 16000, 1411), (2, 4801, 44100, 3000), (2, 9600, 48000, 1536)])
def test_ratio_then_int_node_path(pack, monkeypatch, C, n, sr, kbps):
    """SPEC.md factor_mode "ratio_then_int" + interp "linspace": the output length is int(n * ratio) -- 88187 samples for one second
    of 16 kHz mono at 1411 kbps, no multiple of the input -- through the node arithmetic (PCM_16 hops, autoscale, normalise) against
    oracle.node_run with the matching FatLlamaSpec; a ratio of exactly 1 (48 kHz stereo at 1536 kbps) degenerates to factor 1."""
    import dataclasses
    from egregora_amd import fatllama_engine as fe
    spec = dataclasses.replace(ofl.DEFAULT_SPEC, interp="linspace", factor_mode="ratio_then_int")
    cs = synth(C, n, seed=n, scale=0.4, integer=False)
    want, sr_out = ofl.node_run(cs, sr, 4, 0.6, kbps, True, True, spec=spec)
    monkeypatch.setenv("EGREGORA_FATLLAMA_SPEC", "linspace,ratio_then_int")
    got, sr_dev = fe.node_run(torch.from_numpy(cs), sr, 4, 0.6, kbps, True, True)
    got = got.cpu().numpy()
    assert sr_dev == sr_out and got.shape == want.shape and (want.shape[1] % n != 0 or kbps == 1536)
    lsb = np.abs(got - want) * 32768.0
    assert float(lsb.max()) <= 1.0 + 1e-6 and float(np.mean(lsb > 0.5)) <= 5e-2, (float(lsb.max()), float(np.mean(lsb > 0.5)))


def test_zero_iterations_and_edge_inputs(pack):
    x = synth(2, 64, seed=1)
    want = ofl.enhance_channels(x, 1, 0, 0.6, normalize=False, autoscale=False)
    np.testing.assert_array_equal(run_gpu(pack, x, 1, 0, 0.6), want)
    z = np.zeros((1, 256), np.float32)                      # all-zero input stays zero, no NaN from 0/0
    out = run_gpu(pack, z, 1, 3, 0.6, normalize=True, autoscale=True)
    assert np.all(out == 0)


def test_length_one_is_loud(pack):
    from egregora_amd import fatllama_engine as fe
    with pytest.raises(RuntimeError, match="unsupported|out of range"):
        fe.enhance_device(torch.zeros(1, 1, device="cuda"), 1, 1, 0.6, False, False, False, False)


@pytest.mark.parametrize("normalize,autoscale", [(True, True), (True, False), (False, True), (False, False)])
def test_node_arithmetic_matches_oracle(pack, normalize, autoscale):
    """Unit-scale float in -> PCM_16 hop -> engine -> autoscale/normalise -> write patch -> PCM_16 hop."""
    rng = np.random.Generator(np.random.PCG64(77))
    cs = (synth(2, 24000, seed=9, scale=0.5, integer=False) * np.array([[1.0], [0.4]], np.float32)).astype(np.float32)
    cs[0, 5] = 1.3          # beyond full scale: wraps in the PCM_16 write like libsndfile without clipping
    want, sr_out = ofl.node_run(cs, 48000, 7, 0.6, 1536, normalize, autoscale)
    got = run_gpu(pack, cs, 1, 7, 0.6, normalize, autoscale, pcm_in=True, node_post=True)
    assert sr_out == 48000
    lsb = np.abs(got - want) * 32768.0
    assert float(lsb.max()) <= 1.0 + 1e-6
    assert float(np.mean(lsb > 0.5)) <= 5e-2
    assert om.lsd_audio(want, got)[0] <= 0.05      # post-quantisation LSD, full-band material


def test_full_size_properties_c3_shape(pack):
    """C3 shape (2 x 2,880,000), few iterations: size-independent properties instead of an oracle run.
    (a) idempotence of the loop up to float32 round-off: 1 vs 6 iterations agree to ~1e-5 relative;
    (b) linearity of y + d in a scale factor that is a power of two (bit-exact scaling of every step);
    (c) thr = 0 keeps everything: out == 2*y up to round-off."""
    x = synth(2, 2880000, seed=303)
    a = run_gpu(pack, x, 1, 1, 0.6)
    b = run_gpu(pack, x, 1, 6, 0.6)
    s = float(np.max(np.abs(a)))
    assert float(np.max(np.abs(a - b))) <= 3e-5 * s
    h = run_gpu(pack, (x * np.float32(0.5)).astype(np.float32), 1, 1, 0.3)
    np.testing.assert_array_equal(h * np.float32(2.0), a)
    k = run_gpu(pack, x, 1, 2, 0.0)
    y = x.copy(); y[:, -1] = 0.0
    assert float(np.max(np.abs(k - 2.0 * y))) <= 3e-5 * s


def test_c3_factorisation_with_compile_time_schedules_matches_oracle(pack):
    """N = 2 880 000 (the C3 length: M = 625 x 2304) takes the loop kernels instantiated for fixed radix schedules -- rows 2304 =
    16 x 16 x 9, columns 625 = 25 x 25, butterfly-ordered float stage tables (k_row<., 1>, k_col<., 2>) -- which no smaller test
    length reaches.  One channel, 3 iterations against the oracle and the float64 yardstick; and EGR_FL_SCHED=0 selects the run-time-schedule
    kernels for the same plan."""
    from egregora_amd import fatllama_engine as fe
    info = fe.plan_info(2880000, 1)
    assert (info["M1"], info["M2"]) == (625, 2304)
    x = synth(1, 2880000, seed=2880)
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
    got = run_gpu(pack, x, 1, 3, 0.6)
    scale = float(np.max(np.abs(want)))
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    print(f"\nC3 length, scheduled kernels: max err {float(np.max(np.abs(got - exact))):.3e} (oracle32 {float(np.max(np.abs(want - exact))):.3e}), "
          f"rms {rms(got - exact):.3e} / {rms(want - exact):.3e}, peak {scale:.0f}")
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale
    assert rms(got - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale
    assert om.lsd_audio(want[:, :960000], got[:, :960000])[0] <= 1e-3


def c1_signal():
    """BASELINE configs[0] input (SURVEY 8(d) recipe): 10 s mono 16 kHz, 8 log-spaced sines 80 Hz..6 kHz (1/k) + noise, peak 0.5."""
    n, sr = 160000, 16000
    rng = np.random.Generator(np.random.PCG64(101))
    t = np.arange(n) / sr
    x = sum(np.sin(2 * np.pi * f * t) / (k + 1) for k, f in enumerate(np.geomspace(80.0, 6000.0, 8))) + 0.01 * rng.standard_normal(n)
    return (0.5 * x / np.max(np.abs(x))).astype(np.float32)[None], sr


def test_c1_exactly_as_baseline_states_it_through_the_cpu_node(pack):
    """BASELINE configs[0]: 160 000 samples mono @16 kHz, max_iterations = 50, 1411 kbps (factor 6, N' = 960 000), through
    EgregoraFatLlamaCPU().run (reference egregora_fat_llama_cpu.py:126-134,147: 7 kwargs, upstream's default toggles) against
    oracle.node_run on the same input: PCM_16 output within 1 LSB everywhere, <= 5 % of samples differing; and the raw loop on
    the same data against the float64 yardstick (LSD over the bins float32 can resolve <= 1e-3 dB, see the module docstring; the
    input is eight tones over a -40 dB noise floor up-rated six-fold, so most bins are images far below the tones)."""
    cs, sr = c1_signal()
    node = pack.NODE_CLASS_MAPPINGS["EgregoraFatLlamaCPU"]()
    (res,) = node.run("wav", 50, 0.6, 1411, AUDIO={"waveform": torch.from_numpy(cs)[None], "sample_rate": sr})
    want, sr_out = ofl.node_run(cs, sr, 50, 0.6, 1411, True, True)
    got = res["waveform"][0].numpy()
    assert res["sample_rate"] == sr_out == 96000 and got.shape == want.shape == (1, 960000)
    lsb = np.abs(got - want) * 32768.0
    assert float(lsb.max()) <= 1.0 + 1e-6 and float(np.mean(lsb > 0.5)) <= 5e-2, (float(lsb.max()), float(np.mean(lsb > 0.5)))
    xi = ofl.pcm16_write(cs).astype(np.float32)
    raw = run_gpu(pack, xi, 6, 50, 0.6)
    exact = ofl.enhance_channels(xi, 6, 50, 0.6, normalize=False, autoscale=False, exact=True)
    oracle32 = ofl.enhance_channels(xi, 6, 50, 0.6, normalize=False, autoscale=False)
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    lg, kept = om.lsd_masked(exact, raw, f32_run=oracle32, margin_db=F32_MARGIN_DB)
    lo, _ = om.lsd_masked(exact, oracle32, f32_run=oracle32, margin_db=F32_MARGIN_DB)
    print(f"\nC1 raw loop: rms err device {rms(raw - exact):.3e} / oracle32 {rms(oracle32 - exact):.3e}; LSD over the {kept:.1%} of bins >= "
          f"{F32_MARGIN_DB:.0f} dB above the float32 floor: device {lg:.2e} dB, float32 oracle {lo:.2e} dB; PCM_16 samples differing "
          f"{float(np.mean(lsb > 0.5)):.4f}")
    assert rms(raw - exact) <= 2.5 * rms(oracle32 - exact)
    assert lg <= 1e-3 and kept >= 0.15, (lg, kept)


def test_800_iterations_against_the_oracle(pack):
    """The headline iteration count on 1 s of stereo 48 kHz (N = 48 000 per channel): all 800 iterations on both sides, float64 run
    as the yardstick.  Float32 round-off DOES build up over the iterations in any implementation: with a 0.6 threshold on PCM-scale
    data each iteration is FFT -> IFFT, and element e meets the same rounded twiddle (|W|^2 - 1 ~ 4e-8) every time, a gain that
    compounds to (1 + eps)^800 ~ 3e-5 -- the pocketfft oracle itself ends 2.4e-5 of the peak away from float64.  Gates: the
    device's error against float64 is within 2x (max) / 2.5x (rms) of the float32 oracle's (measured 1.3x / 1.8x, after moving the
    composite twiddles and the 1/M scaling to double precision: before that 2.0x / 2.9x), and the LSD against float64 over the
    bins a float32 transform can resolve (module docstring) is <= 1e-3 dB."""
    x = synth(2, 48000, seed=800)
    want = ofl.enhance_channels(x, 1, 800, 0.6, normalize=False, autoscale=False)
    exact = ofl.enhance_channels(x, 1, 800, 0.6, normalize=False, autoscale=False, exact=True)
    got = run_gpu(pack, x, 1, 800, 0.6)
    scale = float(np.max(np.abs(want)))
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    mg, mo = float(np.max(np.abs(got - exact))), float(np.max(np.abs(want - exact)))
    lg, kept = om.lsd_masked(exact, got, f32_run=want, margin_db=F32_MARGIN_DB)
    lo, _ = om.lsd_masked(exact, want, f32_run=want, margin_db=F32_MARGIN_DB)
    print(f"\n800 iterations: max err device {mg:.3e} oracle32 {mo:.3e} (peak {scale:.0f}); rms {rms(got - exact):.3e} / {rms(want - exact):.3e}; "
          f"LSD vs float64 over the {kept:.1%} of bins >= {F32_MARGIN_DB:.0f} dB above the float32 floor: device {lg:.2e} dB, oracle32 {lo:.2e} dB; "
          f"plain LSD(device, oracle32) {om.lsd_audio(want, got)[0]:.2e} dB")
    assert np.isfinite(got).all()
    assert mg <= 2.0 * mo and mg <= 1e-4 * scale
    assert rms(got - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale
    assert lg <= 1e-3 and kept >= 0.3, (lg, lo, kept)


def f64_loop_on_gpu(x, iters, thr):
    """The oracle's loop (oracle/fatllama.py ist_loop, default spec, factor 1) in FLOAT64 through torch.fft on the GPU: the yardstick
    for lengths at which the host transform is too slow (pocketfft needs ~0.5 s per iteration at N = 2.88 M in float64, 1.5 s for
    N = 2 x a prime).  An independent transform (rocFFT, double precision); pinned to the oracle's own float64 run below."""
    y = torch.from_numpy(np.asarray(x, np.float64)).cuda()
    y[..., -1] = 0.0                               # linear up-rating by 1: the last input sample's slot stays zero (oracle.interpolate)
    t = float(thr)
    d = torch.where(y.abs() > t, y, torch.zeros_like(y))
    n = y.shape[-1]
    for _ in range(int(iters)):
        X = torch.fft.rfft(d, dim=-1)
        X = torch.where(X.abs() > t, X, torch.zeros_like(X))
        d = torch.fft.irfft(X, n=n, dim=-1)
    return (y + d).cpu().numpy()


def test_c3_full_length_800_iterations_against_the_oracle(pack):
    """BASELINE configs[2] at ITS OWN size and iteration count: one channel of 60 s at 48 kHz (N = 2 880 000: the 625 x 2304 plan on
    the two-barrier kernels, whose twiddle runs are specific to that length) and 60 s + 2 samples (no packed plan: the paired chirp-z
    loop), all 800 iterations.  Round-off compounds over the iterations as (1 + eps)^800 in ANY float32 implementation -- at this
    length the float32 pocketfft oracle itself ends 1.6e-4 of the peak from float64 -- so the gates are relative to the oracle's own
    error, as in test_800_iterations_against_the_oracle: the device's error against float64 within 2x (max) / 2.5x (rms) of the
    float32 oracle's, and the LSD against float64 over the bins a float32 transform resolves (module docstring) <= 1e-3 dB; the
    plain LSD (all bins, including those float32 cannot resolve: 0.8e-3 dB for the float32 ORACLE at this length) is printed and
    held to 3x the oracle's, like the rms error.  The float32 oracle runs on the host for N = 2 880 000 (~3 minutes);
    the float64 yardstick is the same loop through torch.fft in double precision on the GPU (checked here against the oracle's own
    float64 run at a small size).  For N + 2 the host transform is Bluestein's (N = 2 x a prime): 248 s for the float32 run, and
    ELEVEN times noisier than at N (max 2.70 = 1.6e-4 of the peak, rms 0.543, plain LSD 1.2e-2 dB against 0.43 / 0.049 / 0.8e-3):
    those numbers were measured once on the GPU box's host by tools/probe_c3_plus2_oracle.py (same data, same seed) and are the
    reference errors the device's chirp-z path is held to at that length; EGR_TEST_SLOW_ORACLE=1 recomputes them here."""
    from egregora_amd import fatllama_engine as fe
    try:
        small = synth(1, 48000, seed=5)
        pin = f64_loop_on_gpu(small, 20, 0.6)
    except Exception as ex:      # noqa: BLE001 -- no double-precision FFT in this torch build: nothing to measure against in bounded time
        pytest.skip(f"torch.fft in float64 on the GPU is not available here ({ex})")
    want_small = ofl.enhance_channels(small, 1, 20, 0.6, normalize=False, autoscale=False, exact=True)
    assert float(np.max(np.abs(pin - want_small))) <= 1e-9 * float(np.max(np.abs(want_small))), "the GPU float64 loop IS the oracle's float64 loop"
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    seg = slice(0, 960000)                         # the reference's metric on the first 20 s (every frame sees the same loop)
    ORACLE32_AT_N_PLUS_2 = oracle32_at_n_plus_2()            # max, rms, plain LSD (dB) vs float64: tools/probe_c3_plus2_oracle.py
    ref_err = None
    for n in (2880000, 2880002):
        info = fe.plan_info(n, 1)
        assert ((info["M1"], info["M2"]) == (625, 2304)) == (n == 2880000) and bool(info.get("chirpz_kind", 0)) == (n != 2880000), info
        x = synth(1, 2880002, seed=2880)[:, :n].copy()
        got = run_gpu(pack, x, 1, 800, 0.6)
        exact = f64_loop_on_gpu(x, 800, 0.6)
        scale = float(np.max(np.abs(exact)))
        mg, rg = float(np.max(np.abs(got - exact))), rms(got - exact)
        slow = bool(int(os.environ.get("EGR_TEST_SLOW_ORACLE", "0")))
        if n != 2880000 and not slow:
            ref_err, lo_plain, want = ORACLE32_AT_N_PLUS_2[:2], ORACLE32_AT_N_PLUS_2[2], None
        else:
            want = ofl.enhance_channels(x, 1, 800, 0.6, normalize=False, autoscale=False)
            ref_err = (float(np.max(np.abs(want - exact))), rms(want - exact))
            lo_plain = om.lsd_audio(exact[:, seg], want[:, seg])[0]
            lo, _ = om.lsd_masked(exact[:, seg], want[:, seg], f32_run=want[:, seg], margin_db=F32_MARGIN_DB)
            print(f"\nfloat32 oracle at N = {n}, 800 iterations: max err {ref_err[0]:.3e} ({ref_err[0] / scale:.2e} of the peak {scale:.0f}), rms {ref_err[1]:.3e}, "
                  f"LSD vs float64 plain {lo_plain:.2e} dB, over the resolvable bins {lo:.2e} dB")
        f32_run = want[:, seg] if want is not None else got[:, seg]  # (N + 2: the device's own deviation sets the floor; `kept` guards it)
        lg, kept = om.lsd_masked(exact[:, seg], got[:, seg], f32_run=f32_run, margin_db=F32_MARGIN_DB)
        lg_plain = om.lsd_audio(exact[:, seg], got[:, seg])[0]
        print(f"device at N = {n} ({'625 x 2304 plan' if n == 2880000 else 'paired chirp-z'}), 800 iterations: max err {mg:.3e} ({mg / scale:.2e} of the peak), "
              f"rms {rg:.3e}; ratios to the float32 oracle {mg / ref_err[0]:.2f} / {rg / ref_err[1]:.2f}; LSD vs float64 plain {lg_plain:.2e} dB, over the "
              f"{kept:.1%} of bins >= {F32_MARGIN_DB:.0f} dB above the float32 floor {lg:.2e} dB")
        assert np.isfinite(got).all()
        # round 6: at the packed length the loop kernels carry their twiddles and butterfly constants as two floats /
        # double runs (csrc/egr_fatllama_wl.h EGR_WL_HILO) and sit BELOW the float32 oracle's own error (measured 0.59x max, 0.86x rms,
        # plain LSD 7.4e-4 dB; round 5: 1.51x / 2.27x / 2.2e-3 dB against gates of 2x / 2.5x / 3x the oracle's LSD)
        # (round 6, chirp-z: butterfly constants as two floats and the four-step twiddle products in double -- max 1.00 / rms 0.193 against the Bluestein
        # oracle's 2.70 / 0.543, plain LSD 4.2e-3 dB against its 1.2e-2: the gates, 2x / 2.5x / 1x the oracle's until round 5, now sit BELOW the oracle)
        kmax, krms = (1.5, 1.5) if n == 2880000 else (0.75, 0.75)
        assert mg <= kmax * ref_err[0] and mg <= 5e-4 * scale, (n, mg, ref_err, scale)
        assert rg <= krms * ref_err[1] + 1e-9 * scale, (n, rg, ref_err)
        if n == 2880000:
            assert lg_plain <= 1e-3, (n, lg_plain)          # north_star's bar on the PLAIN metric, every bin
        # (at N + 2 the Bluestein round-off floor of ANY float32 run leaves only ~2 % of the bins 80 dB above it -- the float32 oracle's
        # plain LSD is 1.2e-2 dB there; the device's chirp-z path must be no worse than the oracle and meet 1e-3 dB on what is resolvable)
        assert lg <= 1e-3 and kept >= (0.3 if n == 2880000 else 0.01), (n, lg, kept)
        assert lg_plain <= (3.0 if n == 2880000 else 0.6) * lo_plain, (n, lg_plain, lo_plain)


def test_60_s_plus_1_sample_stereo_800_iterations_against_float64(pack):
    """The other half of real files: an ODD length (N = 2 880 001: both channels in ONE chirp-z state as a channel pair, rows of 8192
    points) at the headline iteration count, against the float64 loop on the GPU.  The host transform at this length is Bluestein's
    (as at N + 2 in test_c3_full_length_800_iterations_against_the_oracle, whose recorded float32-oracle errors -- max 2.70, rms
    0.543, plain LSD 1.2e-2 dB at the same data scale -- are the reference here too: same class of transform, same round-off floor)."""
    from egregora_amd import fatllama_engine as fe
    n = 2880001
    info = fe.plan_info(n, 1)
    assert info.get("chirpz_kind", 0) == 2, info
    x = synth(2, n, seed=2881)
    try:
        exact = f64_loop_on_gpu(x, 800, 0.6)
    except Exception as ex:      # noqa: BLE001
        pytest.skip(f"torch.fft in float64 on the GPU is not available here ({ex})")
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    got = run_gpu(pack, x, 1, 800, 0.6)
    scale = float(np.max(np.abs(exact)))
    mg, rg = float(np.max(np.abs(got - exact))), rms(got - exact)
    seg = slice(0, 960000)
    lg, kept = om.lsd_masked(exact[:, seg], got[:, seg], f32_run=got[:, seg], margin_db=F32_MARGIN_DB)
    lg_plain = om.lsd_audio(exact[:, seg], got[:, seg])[0]
    print(f"\nN = {n} stereo (channel pair), 800 iterations: max err {mg:.3e} ({mg / scale:.2e} of the peak {scale:.0f}), rms {rg:.3e}; "
          f"LSD vs float64 plain {lg_plain:.2e} dB, over the {kept:.1%} resolvable bins {lg:.2e} dB")
    assert np.isfinite(got).all()
    o_max, o_rms, o_lsd = oracle32_at_n_plus_2()
    # (round 6: measured max 1.17 / rms 0.220 / plain LSD 5.6e-3 dB -- under half the float32 oracle's own errors at N + 2; gates were 2x / 2.5x / 1x the oracle's)
    assert mg <= 0.75 * o_max and mg <= 5e-4 * scale and rg <= 0.75 * o_rms, (mg, rg, scale)
    assert lg <= 1e-3 and kept >= 0.01 and lg_plain <= 0.6 * o_lsd, (lg, kept, lg_plain)


@pytest.mark.parametrize("n,iters,plan", [(2646000, 800, (441, 3000, 1)), (1323000, 800, (441, 1500, 1)), (5760000, 400, (625, 4608, 1)),
                                          (7200000, 200, (625, 2, 2880)), (5292000, 200, (441, 2, 3000)),
                                          (16257024, 50, (441, 4, 4608))])
def test_round_4_plans_at_full_length_against_float64(pack, n, iters, plan):
    """The plans round 4 added, at full length and hundreds of iterations (their twiddle runs are new, so the compounding is
    checked per plan): 60 s at 44.1 kHz (columns on k_col_wl<21, 12>, rows on k_row_wl<30, 10>) and 30 s (odd cross radix 15) at
    the headline count; 120 s at 48 kHz (rows of 4608); 150 s at 48 kHz and 120 s at 44.1 kHz (three levels around the two-barrier
    kernels).  Yardstick: the float64 loop on the GPU (pinned to the oracle's float64 run in
    test_c3_full_length_800_iterations_against_the_oracle); reference error: the stage-by-stage kernels on the same plan
    (EGR_FL_WL=0), which the other tests hold to the float32 oracle."""
    from egregora_amd import fatllama_engine as fe
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"], info["M3"]) == plan, info
    x = synth(1, n, seed=n % 1009)
    try:
        exact = f64_loop_on_gpu(x, iters, 0.6)
    except Exception as ex:      # noqa: BLE001
        pytest.skip(f"torch.fft in float64 on the GPU is not available here ({ex})")
    rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    got = run_gpu(pack, x, 1, iters, 0.6)
    old = os.environ.get("EGR_FL_WL")
    os.environ["EGR_FL_WL"] = "0"
    try:
        fe.release_plans()
        stage = run_gpu(pack, x, 1, iters, 0.6)
        fe.release_plans()
    finally:
        if old is None:
            os.environ.pop("EGR_FL_WL", None)
        else:
            os.environ["EGR_FL_WL"] = old
    scale = float(np.max(np.abs(exact)))
    mg, rg, ms, rs = float(np.max(np.abs(got - exact))), rms(got - exact), float(np.max(np.abs(stage - exact))), rms(stage - exact)
    seg = slice(0, 882000)
    lg, kept = om.lsd_masked(exact[:, seg], got[:, seg], f32_run=stage[:, seg], margin_db=F32_MARGIN_DB)
    print(f"\n{plan}, {iters} iterations: two-barrier kernels max err {mg:.3e} ({mg / scale:.2e} of the peak), rms {rg:.3e}; stage-by-stage "
          f"kernels {ms:.3e} / {rs:.3e}; LSD vs float64 over the {kept:.1%} resolvable bins {lg:.2e} dB")
    assert np.isfinite(got).all()
    assert mg <= 2.0 * ms and mg <= 5e-4 * scale and rg <= 1.5 * rs + 1e-9 * scale, (mg, ms, rg, rs, scale)
    assert lg <= 1e-3 and kept >= 0.15, (lg, kept)       # (the share of bins 80 dB above the float32 floor falls with the length: 35 % at 60 s, 20 % at 30 s)


VARIANTS = [  # (variant list for the device, FatLlamaSpec overrides, threshold, data scale)
    ("", {}, 50.0, 100.0),
    ("soft", {"threshold_kind": "soft"}, 50.0, 100.0),
    ("relative", {"threshold_ref": "relative_to_max"}, 0.02, 8000.0),
    ("relative,soft", {"threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02, 8000.0),
    ("relative,no_init_thr", {"threshold_ref": "relative_to_max", "init_threshold": "none"}, 0.3, 8000.0),
    ("relative,soft,zero_stuff", {"threshold_ref": "relative_to_max", "threshold_kind": "soft", "interp": "zero_stuff"}, 0.05, 8000.0),
    ("zero_stuff,no_init_thr", {"interp": "zero_stuff", "init_threshold": "none"}, 400.0, 100.0),
    ("linspace", {"interp": "linspace"}, 50.0, 100.0),
    ("linspace,relative,soft", {"interp": "linspace", "threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02, 8000.0),
]


@pytest.mark.parametrize("variant,over,thr,scale", VARIANTS)
@pytest.mark.parametrize("C,n,f,iters", [(2, 4800, 1, 4), (1, 3000, 3, 6), (2, 48000, 2, 60), (2, 4801, 1, 4), (1, 1013, 3, 5), (2, 2 * 1013, 1, 4)])
def test_threshold_and_interpolation_variants_match_the_oracle(pack, variant, over, thr, scale, C, n, f, iters):
    """SPEC.md section 3: absolute / relative-to-maximum level x hard / soft shrink, with or without the time-domain pre-threshold,
    linear or zero-insertion up-rating -- each against the oracle run with the matching FatLlamaSpec, at thresholds that really
    gate bins; the last three lengths (odd: channel pairs; prime factor 1013; even with N/2 prime: even/odd packing) run the paired chirp-z path.  A hard threshold may flip a borderline bin, so the bar is energy-relative (1e-6 of the output energy, as in
    test_large_threshold_actually_gates_bins); the soft shrink is continuous and is also held to 2e-5 of the peak."""
    import dataclasses
    spec = dataclasses.replace(ofl.DEFAULT_SPEC, **over)
    x = synth(C, n, seed=n + f + len(variant), scale=scale)
    want = ofl.enhance_channels(x, f, iters, thr, normalize=False, autoscale=False, spec=spec)
    got = run_gpu(pack, x, f, iters, thr, variant=variant)
    assert got.shape == want.shape
    y = np.stack([ofl.interpolate(x[c], f, spec) for c in range(C)])
    d_want, d_got = want - y, got - y
    kept = float(np.sum(d_want.astype(np.float64) ** 2) / max(float(np.sum(y.astype(np.float64) ** 2)), 1e-30))
    assert 1e-4 < kept < 0.9999 or variant == "", (variant, kept)      # the threshold removes a real share of the energy
    num = float(np.sum((got - want).astype(np.float64) ** 2)); den = float(np.sum(want.astype(np.float64) ** 2)) + 1e-30
    assert num / den < 1e-6, (variant, num / den)
    if "soft" in variant:
        assert float(np.max(np.abs(got - want))) <= 2e-5 * float(np.max(np.abs(want))), variant


@pytest.mark.parametrize("variant,over,thr,scale,n", [
    ("relative,soft", {"threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02, 8000.0, 2880000),
    ("", {}, 50.0, 100.0, 2880000),
    ("relative,soft", {"threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02, 8000.0, 2880002),
    # two of the plans round 4 added, against the ORACLE (they were gated against the device's own stage-by-stage kernels and a
    # float64 loop only): 60 s at 44.1 kHz (441 x 3000: k_col_wl<21, 12> + k_row_wl<30, 10>) and 150 s at 48 kHz (three levels, 625 x 2 x 2880)
    ("relative,soft", {"threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02, 8000.0, 2646000),
    ("", {}, 50.0, 100.0, 7200000),
])
def test_real_gating_at_the_headline_length_against_the_oracle(pack, variant, over, thr, scale, n):
    """the default spec's 0.6 absolute threshold gates nothing on int16-scale data, so the full-length parity tests verify
    an FFT -> IFFT identity chain; thresholds that really gate were only tested up to 48 000 samples.  Here BASELINE configs[2]'s own
    length (60 s at 48 kHz, one channel; and 60 s + 2 samples: the paired chirp-z path) runs 50 iterations with a threshold that
    removes a real share of the spectrum every iteration -- soft shrink relative to the maximum, and a hard absolute level -- against
    oracle/fatllama.py with the matching FatLlamaSpec (float32 pocketfft).  Bars as in the small-size variant tests: 1e-6 of the output
    energy (a hard threshold may flip a borderline bin), the continuous soft shrink also 2e-5 of the peak."""
    import dataclasses
    from egregora_amd import fatllama_engine as fe
    spec = dataclasses.replace(ofl.DEFAULT_SPEC, **over)
    info = fe.plan_info(n, 1)
    assert bool(info.get("chirpz_kind", 0)) == (n == 2880002), info
    x = synth(1, n, seed=77 + len(variant), scale=scale)
    want = ofl.enhance_channels(x, 1, 50, thr, normalize=False, autoscale=False, spec=spec)
    got = run_gpu(pack, x, 1, 50, thr, variant=variant)
    assert got.shape == want.shape and np.isfinite(got).all()
    d_want = want - x
    kept = float(np.sum(d_want.astype(np.float64) ** 2) / float(np.sum(x.astype(np.float64) ** 2)))
    num = float(np.sum((got - want).astype(np.float64) ** 2)); den = float(np.sum(want.astype(np.float64) ** 2)) + 1e-30
    peak = float(np.max(np.abs(want)))
    mx = float(np.max(np.abs(got - want)))
    print(f"\nN = {n}, 50 iterations, variant {variant or 'default'!r} thr {thr}: the loop's output carries {kept:.3%} of the input energy on top of the input; "
          f"device vs float32 oracle: energy-relative {num / den:.2e}, max {mx:.3e} ({mx / peak:.2e} of the peak {peak:.0f}); plan {info['M1']} x {info['M2']} x {info['M3']}")
    assert 1e-4 < kept < 0.9999, kept                       # the threshold removes / keeps a real share of the energy
    assert num / den < 1e-6, (variant, num / den)
    if "soft" in variant:
        assert mx <= 2e-5 * peak, (variant, mx, peak)


@pytest.mark.parametrize("C,n,f,iters,split", [
    (2, 48000, 1, 130, None),             # packed two-level plan, two pipelines: five replays of the captured 25-iteration graph + remainder
    (1, 9600, 2, 61, None),               # one pipeline, up-rated
    (2, 2 * 16 * 15 * 64, 1, 57, (16, 15, 64)),      # three levels
    (2, 2880000, 1, 60, None),            # the headline plan (625 x 2304: k_row_wl<16, 12, 1>), ring of 25 slots wrapped twice
    (2, 2 * 1013 * 24, 1, 80, None),      # paired chirp-z, kind 1, two states: graph replay with iteration 0 outside the graph
    (4, 4801, 1, 80, None),               # paired chirp-z, kind 2 (channel pairs share a state: two maxima per workgroup)
    (2, 4801, 1, 30, None),               # kind 2, one state
])
@pytest.mark.parametrize("variant,over,thr", [
    ("relative,soft", {"threshold_ref": "relative_to_max", "threshold_kind": "soft"}, 0.02),
    ("relative", {"threshold_ref": "relative_to_max"}, 0.05),
])
def test_carried_spectrum_maximum_equals_the_recomputed_one(pack, C, n, f, iters, split, variant, over, thr):
    """Round 6: the relative-to-maximum level no longer costs a read-only pass per iteration.  The shrink keeps the
    Hermitian symmetry of the spectrum, so fft(real(ifft(S(X)))) = S(X): the maximum iteration i + 1 will find is the maximum of
    iteration i's shrunk spectrum, which the hook holds in registers and leaves in a ring of slots (one atomic per workgroup); only
    iteration 0 runs a maximum pass.  Held here (a) to the per-iteration pass of rounds 1-5 (variant "recompute"): the level then
    differs by the round-off of one float32 transform pair, 1e-7 relative -- continuous soft shrink: <= 5e-6 of the peak; hard
    threshold: a borderline bin may flip, 1e-6 of the output energy; (b) to its own plain-launch run bit for bit (the ring is
    addressed by iteration mod 25 inside the captured graph); (c) where the oracle finishes in seconds, to oracle/fatllama.py with
    the matching FatLlamaSpec (which recomputes max |X| every iteration, SPEC.md section 3) at the variant tests' bars."""
    import dataclasses
    x = synth(C, n, seed=n + iters + len(variant), scale=8000.0)
    got = run_gpu(pack, x, f, iters, thr, variant=variant, split=split)
    ref = run_gpu(pack, x, f, iters, thr, variant=variant + ",recompute", split=split)
    plain = run_gpu(pack, x, f, iters, thr, variant=variant, split=split, profile=True)
    np.testing.assert_array_equal(got, plain)
    assert np.isfinite(got).all()
    peak = float(np.max(np.abs(ref)))
    num = float(np.sum((got - ref).astype(np.float64) ** 2)); den = float(np.sum(ref.astype(np.float64) ** 2)) + 1e-30
    mx = float(np.max(np.abs(got - ref)))
    print(f"\n{variant} C={C} n={n} f={f} iters={iters}: carried vs recomputed maximum: energy-relative {num / den:.2e}, max {mx / peak:.2e} of the peak")
    assert num / den < 1e-6, (variant, num / den)
    if "soft" in variant:
        assert mx <= 5e-6 * peak, (mx, peak)
    if n * f <= 100000:
        spec = dataclasses.replace(ofl.DEFAULT_SPEC, **over)
        want = ofl.enhance_channels(x, f, iters, thr, normalize=False, autoscale=False, spec=spec)
        y = np.stack([ofl.interpolate(x[c], f, spec) for c in range(C)])
        kept = float(np.sum((want - y).astype(np.float64) ** 2) / float(np.sum(y.astype(np.float64) ** 2)))
        num = float(np.sum((got - want).astype(np.float64) ** 2)); den = float(np.sum(want.astype(np.float64) ** 2)) + 1e-30
        print(f"   vs the float32 oracle: energy-relative {num / den:.2e}; the loop adds {kept:.3%} of the input's energy")
        assert kept > 1e-5 and num / den < 1e-6, (variant, kept, num / den)
        if "soft" in variant:
            assert float(np.max(np.abs(got - want))) <= 2e-5 * float(np.max(np.abs(want)))


@pytest.mark.parametrize("iters", [1, 2, 3, 24, 25, 26, 27, 50, 51, 52, 53, 76, 77, 101])
@pytest.mark.parametrize("C,n", [(2, 9600), (2, 2 * 1013 * 4), (4, 4801)])
def test_carried_maximum_at_every_graph_and_ring_boundary(pack, C, n, iters):
    """The ring of 25 maximum slots and the captured 25-iteration graph meet at iteration counts around multiples of 25 (the packed loop replays
    (iters - 1) // 25 graphs once iters > 50 and runs the rest as plain launches; the chirp-z loop keeps iteration 0 outside the graph): every
    count from 1 to 101 that sits on such an edge, soft shrink relative to the maximum (the level changes every iteration: a slot read one
    iteration early or late shows), against the maximum pass per iteration."""
    x = synth(C, n, seed=n + iters, scale=8000.0)
    got = run_gpu(pack, x, 1, iters, 0.03, variant="relative,soft")
    ref = run_gpu(pack, x, 1, iters, 0.03, variant="relative,soft,recompute")
    peak = float(np.max(np.abs(ref)))
    assert np.isfinite(got).all() and float(np.max(np.abs(got - ref))) <= 5e-6 * peak, (C, n, iters, float(np.max(np.abs(got - ref))) / peak)
