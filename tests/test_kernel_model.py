"""CPU check of the engine's index math (tests/kernel_model.py) against numpy.fft."""
import numpy as np
import pytest

import kernel_model as km


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6, 7, 8, 12, 15, 16, 20, 60, 75, 128, 1200, 1440, 11 * 13 * 2])
def test_stockham_stage_addressing(L):
    rng = np.random.default_rng(L)
    x = rng.standard_normal((3, L)) + 1j * rng.standard_normal((3, L))
    np.testing.assert_allclose(km.stockham_fft(x), np.fft.fft(x, axis=-1), atol=1e-9 * max(1, L))
    np.testing.assert_allclose(km.stockham_fft(x, inverse=True), np.fft.ifft(x, axis=-1) * L, atol=1e-9 * max(1, L))


@pytest.mark.parametrize("M1,M2", [(1, 8), (1, 7), (2, 6), (3, 5), (4, 4), (4, 5), (5, 4), (6, 9), (8, 15), (16, 12), (12, 35), (1, 1), (2, 1)])
@pytest.mark.parametrize("iters", [1, 3])
def test_ist_matches_fft_loop(M1, M2, iters):
    N = 2 * M1 * M2
    rng = np.random.default_rng(N + iters)
    y = np.rint(rng.standard_normal(N) * 50)
    y[rng.integers(0, N, N // 4)] = 0.0
    thr = 20.0  # large enough that a good share of bins / samples is actually thresholded
    d = np.where(np.abs(y) > thr, y, 0.0)
    for _ in range(iters):
        X = np.fft.fft(d)
        X = np.where(np.abs(X) > thr, X, 0)
        d = np.fft.ifft(X).real
    p = km.Plan(N, M1, M2)
    got = km.ist(p, y, iters, thr)
    np.testing.assert_allclose(got, d, atol=1e-9 * N)


@pytest.mark.parametrize("M1,M2,M3", [(2, 3, 4), (3, 2, 5), (4, 4, 4), (5, 3, 2), (1, 3, 4), (2, 1, 6), (6, 5, 7), (4, 6, 9)])
@pytest.mark.parametrize("iters", [1, 3])
def test_ist_three_level_matches_fft_loop(M1, M2, M3, iters):
    N = 2 * M1 * M2 * M3
    rng = np.random.default_rng(N * 7 + iters)
    y = np.rint(rng.standard_normal(N) * 50)
    y[rng.integers(0, N, N // 4)] = 0.0
    thr = 20.0
    d = np.where(np.abs(y) > thr, y, 0.0)
    for _ in range(iters):
        X = np.fft.fft(d)
        X = np.where(np.abs(X) > thr, X, 0)
        d = np.fft.ifft(X).real
    got = km.ist3(km.Plan3(N, M1, M2, M3), y, iters, thr)
    np.testing.assert_allclose(got, d, atol=1e-9 * N)


@pytest.mark.parametrize("n1,q", [(16, 12), (6, 8), (4, 12), (12, 8), (30, 8), (20, 12), (12, 16), (28, 12), (16, 16), (30, 10), (4, 10), (14, 10), (15, 8), (5, 10), (21, 10), (15, 16)])
def test_two_barrier_row_kernel_lane_map(n1, q):
    """csrc/egr_fatllama_wl.h k_row_wl<N1, Q>: the (block, lane, register) a thread holds is X[k1 + N1 (c + Q d)], the partner L-1-k
    of the real split sits at (N1-1 - k1, Q-1 - c, Q-1 - d) -- the reversed-lane unit of row b -- and the backward steps invert the
    forward; the pair twiddles of a lane's consecutive registers differ by W_(2L)^(N1 Q) = W_(2Q)."""
    L = n1 * q * q
    rng = np.random.default_rng(3)
    x = rng.standard_normal(L) + 1j * rng.standard_normal(L)
    regs = km.wl_row_forward(x, n1, q)
    X = np.fft.fft(x)
    for k1 in range(n1):
        for c in range(q):
            for d in range(q):
                k = k1 + n1 * (c + q * d)
                assert abs(regs[k1, c, d] - X[k]) < 1e-8
                p1, pc, pd = km.wl_row_partner(k1, c, d, n1, q)
                assert p1 + n1 * (pc + q * pd) == L - 1 - k
    np.testing.assert_allclose(km.wl_row_inverse(regs) / L, x, atol=1e-11)
    assert abs(np.exp(-2j * np.pi * n1 * q / (2 * L)) - np.exp(-1j * np.pi / q)) < 1e-15


def test_two_barrier_column_kernel_needs_no_exchange_between_inverse_and_forward():
    rng = np.random.default_rng(4)
    u = rng.standard_normal(625) + 1j * rng.standard_normal(625)
    t, X = km.wl_col_mid(u)
    np.testing.assert_allclose(t, np.fft.ifft(u) * 625, atol=1e-9)
    np.testing.assert_allclose(X, 625 * u, atol=1e-8)


@pytest.mark.parametrize("la,lb", [(2, 1), (7, 1), (16, 1), (2, 7), (3, 5), (4, 5), (6, 10), (8, 9), (8, 16), (12, 12)])
def test_one_barrier_inner_kernel_index_map(la, lb):
    rng = np.random.default_rng(la * 100 + lb)
    x = rng.standard_normal(la * lb) + 1j * rng.standard_normal(la * lb)
    np.testing.assert_allclose(km.wl_inner(x, la, lb, True), np.fft.fft(x), atol=1e-10)
    np.testing.assert_allclose(km.wl_inner(x, la, lb, False), np.fft.ifft(x) * la * lb, atol=1e-10)


def test_planner_prefers_the_two_barrier_kernels_for_whole_minutes_at_48k():
    """N = 2 x 625 x k x 2304 (k >= 3) plans as 625 x k x 2304 (csrc/egr_plan.cpp): BASELINE configs[4]'s 172.8 M samples -> k = 60."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from packload import load_pack
    load_pack()
    from egregora_amd import fatllama_engine as fe
    for n, want in ((2880000, (625, 2304, 1)), (5760000, (625, 4608, 1)), (960000, (625, 768, 1)), (3600000, (625, 2880, 1)), (8640000, (625, 3, 2304)), (28800000, (625, 10, 2304)),
                    (172800000, (625, 60, 2304)), (86400000, (625, 30, 2304))):
        i = fe.plan_info(n, 1)
        assert i["supported"] and (i["M1"], i["M2"], i["M3"]) == want, (n, i)
