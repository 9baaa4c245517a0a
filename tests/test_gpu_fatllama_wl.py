"""The two-barrier loop kernels of the hot Fat-Llama plan (csrc/egr_fatllama_wl.h: k_row_wl for rows of 2304 points, k_col_wl for
outer columns of 625 points) against the stage-by-stage kernels they replace (EGR_FL_WL=0: k_row<false, 1> / k_col<1, 2>) and
against the oracle.  Same state layout and per-element arithmetic, another factorisation order: agreement to float32 round-off.
Upstream call site: /root/reference/egregora_fat_llama_gpu.py:213-224 (feed.upscale; parity unpinned, SPEC.md section 3)."""
import os

import numpy as np
import pytest
import torch

from oracle import fatllama as ofl
from test_gpu_fatllama import synth

pytestmark = pytest.mark.gpu
FLAGS = dict(normalize=False, autoscale=False, pcm_in=False, node_post=False)


def run(x, iters, thr=0.6, wl=True, **kw):
    from egregora_amd import fatllama_engine as fe
    old = os.environ.get("EGR_FL_WL")
    os.environ["EGR_FL_WL"] = "1" if wl else "0"
    try:
        fe.release_plans()                      # the switch is read when a plan is built
        y = fe.enhance_device(torch.from_numpy(x).cuda(), 1, iters, thr, **FLAGS, **kw).cpu().numpy()
        fe.release_plans()
    finally:
        if old is None:
            os.environ.pop("EGR_FL_WL", None)
        else:
            os.environ["EGR_FL_WL"] = old
    return y


def rms(a):
    return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))


@pytest.mark.parametrize("channels,iters", [(1, 1), (2, 3), (3, 2)])
def test_two_barrier_kernels_agree_with_the_stage_by_stage_kernels(pack, channels, iters):
    """C3 length (M = 625 x 2304): every row pair incl. the self-paired row 0, every column tile; 1-3 channels (one and two
    channel pipelines)."""
    x = synth(channels, 2880000, seed=77 + channels)
    a = run(x, iters, wl=True)
    b = run(x, iters, wl=False)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    print(f"\nwl vs stage-by-stage, {channels} ch, {iters} it: max diff {err:.3e} of peak {scale:.0f} ({err / scale:.2e}), rms {rms(a - b) / scale:.2e}")
    assert np.isfinite(a).all()
    assert err <= 4e-6 * scale
    assert rms(a - b) <= 4e-7 * scale


def test_two_barrier_kernels_with_a_threshold_that_gates_a_real_share_of_the_spectrum(pack):
    """thr = 1.5e5 on PCM-scale data zeroes most noise bins (|X| ~ 80 sqrt(N) = 1.4e5; no time-domain pre-threshold, which would
    zero every sample at that level): a hard threshold may flip a borderline bin between two float32 implementations, so the
    bound is on the energy of the difference (as for the SPEC variants)."""
    x = synth(2, 2880000, seed=13)
    a = run(x, 4, thr=1.5e5, wl=True, variant="no_init_thr")
    b = run(x, 4, thr=1.5e5, wl=False, variant="no_init_thr")
    y = x.copy(); y[:, -1] = 0
    d = b - y                                                          # out = y + d: the part the loop adds
    assert rms(d) > 0.5 * rms(y) and rms(d - y) > 1e-3 * rms(y)          # the tones survive, the threshold really removed something
    assert float(np.sum(np.square(a - b, dtype=np.float64))) <= 1e-6 * float(np.sum(np.square(b, dtype=np.float64)))


def test_two_barrier_kernels_against_the_oracle_and_float64(pack):
    x = synth(1, 2880000, seed=5)
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
    got = run(x, 3, wl=True)
    old = run(x, 3, wl=False)
    scale = float(np.max(np.abs(want)))
    print(f"\nC3 length vs float64: two-barrier kernels max {float(np.max(np.abs(got - exact))):.3e} rms {rms(got - exact):.3e}; stage-by-stage "
          f"max {float(np.max(np.abs(old - exact))):.3e} rms {rms(old - exact):.3e}; oracle32 max {float(np.max(np.abs(want - exact))):.3e} rms {rms(want - exact):.3e}")
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale
    assert rms(got - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale
    assert rms(got - exact) <= 1.25 * rms(old - exact) + 1e-9 * scale          # no less accurate than the kernels it replaces


def test_two_barrier_kernels_over_200_iterations(pack):
    """Round-off compounds over the loop: the two kernel families must stay together over a long run."""
    x = synth(2, 2880000, seed=9)
    a = run(x, 200, wl=True)
    b = run(x, 200, wl=False)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    print(f"\n200 iterations: max diff {err / scale:.2e} of the peak, rms {rms(a - b) / scale:.2e}")
    assert err <= 5e-5 * scale and rms(a - b) <= 5e-6 * scale


def test_two_barrier_kernels_inside_a_three_level_plan(pack):
    """M = 625 x 2 x 2304: the outer column pass and the row pass of a three-level plan take the same kernels (state rows are
    addressed through (Ma, Mb), 4608 columns); checked against the planner's own two-level plan for that length and the oracle."""
    n = 2 * 625 * 2 * 2304
    x = synth(1, n, seed=11)
    a = run(x, 2, wl=True, split=(625, 2, 2304))
    b = run(x, 2, wl=True)
    c = run(x, 2, wl=False, split=(625, 2, 2304))
    want = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False)
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(a - c))) <= 4e-6 * scale
    assert float(np.max(np.abs(a - b))) <= 1e-5 * scale
    assert float(np.max(np.abs(a - want))) <= 2e-5 * scale


@pytest.mark.parametrize("cols,k,rows", [(625, 2, 2880), (441, 3, 3000), (625, 13, 384), (441, 4, 2500)])
def test_planner_takes_three_levels_around_any_two_barrier_row_length(pack, cols, k, rows):
    """Files beyond ~100 s: N / 2 = C x k x L with C = 625 or 441 columns on k_col_wl and rows of L points on k_row_wl (150 s at 48 kHz
    = 625 x 2 x 2880, 180 s at 44.1 kHz = 441 x 3 x 3000, ...) -- the planner's own choice when no two-level plan has such columns;
    against the stage-by-stage kernels on the same split and, where the host transform is quick, the oracle."""
    from egregora_amd import fatllama_engine as fe
    n = 2 * cols * k * rows
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"], info["M3"], info["levels"]) == (cols, k, rows, 3), info
    x = synth(1, n, seed=cols + k + rows)
    a = run(x, 2, wl=True)
    c = run(x, 2, wl=False, split=(cols, k, rows))
    scale = float(np.max(np.abs(c)))
    assert np.isfinite(a).all() and float(np.max(np.abs(a - c))) <= 4e-6 * scale and rms(a - c) <= 4e-7 * scale
    if n <= 1500000:
        want = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False)
        assert float(np.max(np.abs(a - want))) <= 2e-5 * scale


INNER = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 22, 24, 25, 26, 27, 28, 30, 32, 33, 35, 36, 39, 40, 42, 44, 45, 48, 49, 50,
         52, 54, 55, 56, 60, 63, 64, 72, 80, 90, 96, 100, 120, 144]


@pytest.mark.parametrize("inner", INNER)
def test_one_barrier_inner_pass_of_three_level_plans(pack, inner):
    """k_colb_wl<LA, LB> serves the inner column pass (length M2) of a three-level plan for every instantiated length: M = 6 x M2 x 50
    (50 columns: ragged last tile for every tile width) against the oracle, and against the stage-by-stage inner kernels."""
    n = 2 * 6 * inner * 50
    x = synth(2, n, seed=1000 + inner)
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
    got = run(x, 3, wl=True, split=(6, inner, 50))
    old = run(x, 3, wl=False, split=(6, inner, 50))
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got - want))) <= 2e-5 * scale
    assert float(np.max(np.abs(got - old))) <= 4e-6 * scale
    assert rms(got - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale



def run_pz(x, iters, wl, thr=0.6, **kw):
    from egregora_amd import fatllama_engine as fe
    old = os.environ.get("EGR_PZ_COLWL")
    os.environ["EGR_PZ_COLWL"] = "1" if wl else "0"
    try:
        fe.release_plans()
        y = fe.enhance_device(torch.from_numpy(x).cuda(), 1, iters, thr, **FLAGS, **kw).cpu().numpy()
        fe.release_plans()
    finally:
        if old is None:
            os.environ.pop("EGR_PZ_COLWL", None)
        else:
            os.environ["EGR_PZ_COLWL"] = old
    return y


@pytest.mark.parametrize("n,channels,kw,thr", [
    (512 * 4096 - 6, 2, {}, 0.6),                      # columns of 512 = 16 x 32
    (720 * 4096 - 6, 1, {}, 0.6),                      # 720 = 24 x 30, one state
    (2880001, 3, {}, 0.6),                             # odd length: channel pairs (kind 2), rows of 8192, the last state one channel
    (600 * 4096 - 6, 2, {"variant": "relative,soft"}, 0.02),     # hook variants run through the same pz_pair_hook
    (1024 * 4096 - 6, 2, {}, 0.6),                     # 1024 = 32 x 32 on rows of 8192
])
def test_column_passes_of_the_chirp_z_path_on_the_thread_per_column_kernels(pack, n, channels, kw, thr):
    """Lengths without a packed plan: the crop pass and (even lengths) the spectrum pass run k_pzcol_wl / k_pzpair_wl
    (csrc/egr_fatllama_pz.hip: 2 / 4 workgroup barriers) for every column length with a compile-time schedule; against the
    stage-by-stage kernels (EGR_PZ_COLWL=0) and, for the shortest, the oracle and float64."""
    from egregora_amd import fatllama_engine as fe
    info = fe.plan_info(n, 1)
    assert info["bluestein"], info
    scale_in = 8000.0
    x = synth(channels, n, seed=n % 1000, scale=scale_in)
    a = run_pz(x, 3, True, thr, **kw)
    b = run_pz(x, 3, False, thr, **kw)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    print(f"\nchirp-z column passes, n = {n} ({info['M1']} x {info['M2']}): max diff {err / scale:.2e} of the peak, rms {rms(a - b) / scale:.2e}")
    assert np.isfinite(a).all()
    if kw:
        assert float(np.sum(np.square(a - b, dtype=np.float64))) <= 1e-6 * float(np.sum(np.square(b, dtype=np.float64)))
    else:
        assert err <= 4e-6 * scale and rms(a - b) <= 4e-7 * scale
    if n == 512 * 4096 - 6:
        want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
        exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
        assert float(np.max(np.abs(a - want))) <= 2e-5 * scale
        assert rms(a - exact) <= 2.5 * rms(want - exact) + 5e-8 * scale


@pytest.mark.parametrize("variant,thr", [("soft", 50.0), ("relative", 0.02), ("relative,soft", 0.02), ("relative,no_init_thr", 0.3)])
def test_hook_variants_on_the_two_barrier_row_kernel(pack, variant, thr):
    """SPEC.md section 3 variants at the C3 length: k_row_wl<1> (level from the iteration's spectrum maximum and / or soft shrink) and
    k_row_wl<2> (the maximum's reduction pass) against the stage-by-stage kernels and the oracle with the matching FatLlamaSpec."""
    scale_in = 100.0 if variant == "soft" else 8000.0
    x = synth(2, 2880000, seed=21, scale=scale_in)
    a = run(x, 3, thr=thr, wl=True, variant=variant)
    b = run(x, 3, thr=thr, wl=False, variant=variant)
    eb = float(np.sum(np.square(b, dtype=np.float64)))
    assert np.isfinite(a).all()
    assert float(np.sum(np.square(a - b, dtype=np.float64))) <= 1e-6 * eb       # a borderline bin may flip between two float32 runs
    spec = ofl.FatLlamaSpec(threshold_ref="relative_to_max" if "relative" in variant else "absolute",
                            threshold_kind="soft" if "soft" in variant else "hard",
                            init_threshold="none" if "no_init_thr" in variant else "same")
    want = ofl.enhance_channels(x, 1, 3, thr, normalize=False, autoscale=False, spec=spec)
    assert float(np.sum(np.square(a - want, dtype=np.float64))) <= 1e-6 * float(np.sum(np.square(want, dtype=np.float64)))
    y = x.copy(); y[:, -1] = 0
    assert rms(a - 2 * y) > 1e-4 * rms(y)                                        # the threshold removed something


@pytest.mark.parametrize("n", [480000, 960000, 3600000])
def test_two_barrier_column_kernel_next_to_the_stage_by_stage_row_kernel(pack, n):
    """The planner takes columns of 625 points whenever N / 2 has the factor (10 s, 20 s, 75 s at 48 kHz: 625 x 384 / 768 / 2880):
    k_col_wl serves the column pass, the run-time-schedule k_row the rows.  Against EGR_FL_WL=0 and the oracle."""
    from egregora_amd import fatllama_engine as fe
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["levels"]) == (625, 2), info
    x = synth(2, n, seed=n % 1013)
    a = run(x, 3, wl=True)
    b = run(x, 3, wl=False)
    scale = float(np.max(np.abs(b)))
    assert float(np.max(np.abs(a - b))) <= 4e-6 * scale and rms(a - b) <= 4e-7 * scale
    if n <= 960000:
        want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
        exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
        assert float(np.max(np.abs(a - want))) <= 2e-5 * scale
        assert rms(a - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale


ROW_LENGTHS = [384, 576, 768, 1152, 1536, 1728, 1920, 2880, 3072, 3456, 4032, 4096, 4608, 960, 1344, 3840]       # the last three: odd cross radix


@pytest.mark.parametrize("rows", ROW_LENGTHS)
def test_two_barrier_row_kernel_for_every_instantiated_row_length(pack, rows):
    """k_row_wl<N1, Q> (rows of N1 Q^2 points) next to k_col_wl: M = 625 x rows -- 10 s ... 106 s at 48 kHz.  Against the
    stage-by-stage kernels and, for the shorter ones, the oracle and float64."""
    from egregora_amd import fatllama_engine as fe
    n = 2 * 625 * rows
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"], info["levels"]) == (625, rows, 2), info
    x = synth(2, n, seed=rows)
    a = run(x, 3, wl=True)
    b = run(x, 3, wl=False)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    print(f"\nrows of {rows}: max diff {err / scale:.2e} of the peak, rms {rms(a - b) / scale:.2e}")
    assert np.isfinite(a).all() and err <= 4e-6 * scale and rms(a - b) <= 4e-7 * scale
    if n <= 1500000:
        want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
        exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
        assert float(np.max(np.abs(a - want))) <= 2e-5 * scale
        assert rms(a - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale


ROW_LENGTHS_441 = [400, 600, 800, 1000, 1200, 1400, 1600, 1800, 2000, 2400, 2800, 3000, 3200, 300, 500, 700, 900, 1100, 1300, 1500, 2100, 2500]      # the last nine: odd cross radix


@pytest.mark.parametrize("rows", ROW_LENGTHS_441)
def test_44k1_family_columns_of_441_and_rows_of_50_t(pack, rows):
    """T seconds at 44.1 kHz: N / 2 = 22 050 T = 441 x 50 T.  The planner takes columns of 441 = 21 x 21 points (k_col_wl<21, 12>)
    and rows of 50 T = (T / 2) x 10^2 points (k_row_wl<T / 2, 10>, T = 8 ... 64 s) -- against the stage-by-stage kernels the same
    plan runs with EGR_FL_WL=0 and, for the shorter ones, the oracle and float64."""
    from egregora_amd import fatllama_engine as fe
    n = 2 * 441 * rows
    assert n == 44100 * (rows // 50)
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"], info["levels"]) == (441, rows, 2), info
    x = synth(2, n, seed=rows)
    a = run(x, 3, wl=True)
    b = run(x, 3, wl=False)
    scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    print(f"\n441 x {rows}: max diff {err / scale:.2e} of the peak, rms {rms(a - b) / scale:.2e}")
    assert np.isfinite(a).all() and err <= 4e-6 * scale and rms(a - b) <= 4e-7 * scale
    if n <= 1500000:
        want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
        exact = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False, exact=True)
        assert float(np.max(np.abs(a - want))) <= 2e-5 * scale
        assert rms(a - exact) <= 2.5 * rms(want - exact) + 1e-9 * scale


@pytest.mark.parametrize("rows,channels", [(2600, 2), (2200, 1), (50, 3)])
def test_columns_of_441_next_to_the_stage_by_stage_row_kernel(pack, rows, channels):
    """k_col_wl<21, 12> with row lengths that have no k_row_wl instantiation (2600 and 2200 are 26 x 100 and 22 x 100: no register butterfly of that radix; 50 leaves a RAGGED last tile: 50 = 4 x 12 + 2 columns), explicit split."""
    n = 2 * 441 * rows
    x = synth(channels, n, seed=rows + 1)
    a = run(x, 3, wl=True, split=(441, rows, 1))
    b = run(x, 3, wl=False, split=(441, rows, 1))
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(a - b))) <= 4e-6 * scale and rms(a - b) <= 4e-7 * scale
    assert float(np.max(np.abs(a - want))) <= 2e-5 * scale


@pytest.mark.parametrize("rows", [960, 1500])
def test_odd_cross_radix_with_an_even_row_count(pack, rows):
    """k_row_wl<15, 8> / <15, 10> with 320 rows: two self-paired rows (o = 0 and o = 160), whose middle block (7 of 15) is its own
    partner -- the unit transforms it once and parks its second half on the spare LDS block."""
    n = 2 * 320 * rows
    x = synth(2, n, seed=rows + 7)
    a = run(x, 3, wl=True, split=(320, rows, 1))
    b = run(x, 3, wl=False, split=(320, rows, 1))
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    scale = float(np.max(np.abs(want)))
    assert np.isfinite(a).all()
    assert float(np.max(np.abs(a - b))) <= 4e-6 * scale
    assert float(np.max(np.abs(a - want))) <= 2e-5 * scale


def test_two_barrier_row_kernel_with_an_even_row_count(pack):
    """M = 320 x 768: an even number of rows has TWO self-paired rows (o = 0 and o = R / 2), both on the LDS hook path of k_row_wl;
    the columns run the run-time-schedule kernel."""
    n = 2 * 320 * 768
    x = synth(2, n, seed=320)
    a = run(x, 3, wl=True, split=(320, 768, 1))
    b = run(x, 3, wl=False, split=(320, 768, 1))
    want = ofl.enhance_channels(x, 1, 3, 0.6, normalize=False, autoscale=False)
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(a - b))) <= 4e-6 * scale
    assert float(np.max(np.abs(a - want))) <= 2e-5 * scale


@pytest.mark.parametrize("rows,variant,thr", [(768, "relative,soft", 0.02), (2880, "soft", 50.0), (4608, "relative", 0.02)])
def test_hook_variants_on_other_row_lengths(pack, rows, variant, thr):
    """k_row_wl<N1, Q, 1> / <N1, Q, 2> for row lengths other than the C3 plan's, against the stage-by-stage kernels."""
    n = 2 * 625 * rows
    x = synth(2, n, seed=rows + 1, scale=100.0 if variant == "soft" else 8000.0)
    a = run(x, 3, thr=thr, wl=True, variant=variant)
    b = run(x, 3, thr=thr, wl=False, variant=variant)
    assert np.isfinite(a).all()
    assert float(np.sum(np.square(a - b, dtype=np.float64))) <= 1e-6 * float(np.sum(np.square(b, dtype=np.float64)))
    y = x.copy(); y[:, -1] = 0
    assert rms(a - 2 * y) > 1e-4 * rms(y)


@pytest.mark.parametrize("rows,variant,thr", [(3000, "relative,soft", 0.02), (1500, "soft", 50.0), (500, "relative", 0.02)])
def test_hook_variants_next_to_columns_of_441(pack, rows, variant, thr):
    """The SPEC.md section 3 variants on the 44.1 kHz plans: k_row_wl<N1, 10, 1> / <N1, 10, 2> (N1 = 30 even, 15 and 5 odd -- the
    self-paired row of an odd cross radix goes through the LDS hook path with its spare block) against the stage-by-stage kernels."""
    from egregora_amd import fatllama_engine as fe
    n = 2 * 441 * rows
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"]) == (441, rows), info
    x = synth(2, n, seed=rows + 3, scale=100.0 if variant == "soft" else 8000.0)
    a = run(x, 3, thr=thr, wl=True, variant=variant)
    b = run(x, 3, thr=thr, wl=False, variant=variant)
    assert np.isfinite(a).all()
    assert float(np.sum(np.square(a - b, dtype=np.float64))) <= 1e-6 * float(np.sum(np.square(b, dtype=np.float64)))
    y = x.copy(); y[:, -1] = 0
    assert rms(a - 2 * y) > 1e-4 * rms(y)


def test_120s_plan_625x4608_on_the_generic_row_kernel_and_the_filters(pack):
    """N = 5 760 000 (120 s at 48 kHz) plans as 625 x 4608 -- a row longer than the generic kernels' 4096-point limit, admitted by
    the planner for k_row_wl<32, 12> (csrc/egr_plan.cpp).  The same plan also serves EGR_FL_WL=0 and the single-pass users of the
    plan (egr_band_filter / egr_spectral_gain through k_row<false> with 147 KB of LDS): those paths must work at that row length
    too.  Loop: two-barrier kernels vs stage-by-stage kernels vs the oracle; band filter: Parseval against numpy's rfft."""
    from egregora_amd import device_ops as ops, fatllama_engine as fe
    n = 5760000
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"]) == (625, 4608), info
    x = synth(1, n, seed=120)
    a = run(x, 2, wl=True)
    b = run(x, 2, wl=False)
    want = ofl.enhance_channels(x, 1, 2, 0.6, normalize=False, autoscale=False)
    scale = float(np.max(np.abs(want)))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert float(np.max(np.abs(a - want))) <= 2e-5 * scale and float(np.max(np.abs(b - want))) <= 2e-5 * scale
    assert float(np.max(np.abs(a - b))) <= 4e-6 * scale
    # the high-band energy ratio of the null-test node on the same plan (forward passes + gain hook + inverse passes)
    xt = torch.from_numpy((x / 32768.0).astype(np.float32)).cuda()
    got = ops.band_energy_hi_db(xt, 48000, 8000.0)
    X = np.abs(np.fft.rfft(x[0].astype(np.float64) / 32768.0)) ** 2
    f = np.fft.rfftfreq(n, 1.0 / 48000)
    ref = 10.0 * np.log10(X[f >= 8000.0].sum() / (X.sum() + 1e-20) + 1e-20)
    assert abs(got - ref) <= 1e-3, (got, ref)


@pytest.mark.parametrize("n,plan", [(7200000, (625, 2, 2880)), (5292000, (441, 2, 3000))])
def test_single_pass_users_of_the_new_three_level_plans(pack, n, plan):
    """The plans the round-4 planner takes for files beyond ~100 s (150 s at 48 kHz, 120 s at 44.1 kHz) also serve the single-pass
    users of a plan (egr_band_filter / egr_spectral_gain: forward passes + gain hook + inverse passes on the stage-by-stage
    kernels): the high-band energy ratio of the null-test node against numpy's rfft, on the planner's own plan."""
    from egregora_amd import device_ops as ops, fatllama_engine as fe
    info = fe.plan_info(n, 1)
    assert (info["M1"], info["M2"], info["M3"]) == plan, info
    x = synth(1, n, seed=n % 977)
    xt = torch.from_numpy((x / 32768.0).astype(np.float32)).cuda()
    got = ops.band_energy_hi_db(xt, 48000, 8000.0)
    X = np.abs(np.fft.rfft(x[0].astype(np.float64) / 32768.0)) ** 2
    f = np.fft.rfftfreq(n, 1.0 / 48000)
    ref = 10.0 * np.log10(X[f >= 8000.0].sum() / (X.sum() + 1e-20) + 1e-20)
    assert abs(got - ref) <= 1e-3, (got, ref)
