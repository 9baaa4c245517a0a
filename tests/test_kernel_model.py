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
