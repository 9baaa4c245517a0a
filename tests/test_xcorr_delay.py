"""GCC-PHAT delay estimator of the null-test suite on the Fat-Llama transform passes, against
fixture G12 captured from the reference's _xcorr_delay (tests/golden/make_golden_xcorr.py).

not gpu : the oracle restatement equals the reference bit for bit (numpy float32 transforms on both sides), including quirk Q8
          (lag 0 sits one index left of the search centre: the result is the true lag minus one)
gpu     : device_ops.xcorr_delay (two-for-one complex transform of a + i b, PHAT hook in the row pass, inverse passes, peak
          kernel) within 5e-3 samples of the reference on integer and fractional delays up to 150 000-sample signals
"""
import numpy as np
import pytest
import torch

from conftest import gjson


def cases():
    rng = np.random.Generator(np.random.PCG64(12))
    out = []
    for name, n, d, ms in (("int+3", 6000, 3.0, 200), ("int-7", 6000, -7.0, 200), ("zero", 5000, 0.0, 64), ("frac+2.4", 9000, 2.4, 300),
                           ("frac-11.75", 20000, -11.75, 480), ("long+40", 150000, 40.0, 4800)):
        a = rng.standard_normal(n).astype(np.float32)
        k = np.arange(-32, 33)
        h = np.sinc(k - d + np.round(d)) * np.hanning(65)
        b = np.convolve(np.roll(a, int(np.round(d))), h, mode="same") + 0.01 * rng.standard_normal(n)
        out.append((name, a, b.astype(np.float32), ms))
    return out


def test_oracle_equals_reference():
    from oracle import metrics as om
    g = gjson("g12_xcorr")
    for name, a, b, ms in cases():
        assert om.xcorr_delay(a, b, 48000, ms) == g[name]["delay"], name
    assert abs(g["zero"]["delay"] + 1.0) < 1e-3            # quirk Q8


@pytest.mark.gpu
def test_device_delay_matches_reference(pack):
    from egregora_amd import device_ops
    g = gjson("g12_xcorr")
    for name, a, b, ms in cases():
        got = device_ops.xcorr_delay(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 48000, ms)
        assert abs(got - g[name]["delay"]) <= 5e-3, (name, got, g[name]["delay"])
    with pytest.raises(RuntimeError, match="max_shift"):
        device_ops.xcorr_delay(torch.zeros(100).cuda(), torch.zeros(100).cuda(), 48000, 200)
