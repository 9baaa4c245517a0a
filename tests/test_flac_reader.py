"""The FLAC reader behind the Fat-Llama nodes' audio_path / audio_url inputs (reference
egregora_fat_llama_gpu.py:61-78 reads them with libsndfile): bit-exact decode of streams produced by the test-only encoder
(tests/flac_encoder.py) across every subframe type, residual coding, stereo mode and header variant; float conversion as
sf.read(dtype="float32") (int / 2**(bits-1)); CRC and sync errors are loud."""
import numpy as np
import pytest

from flac_encoder import encode


def pcm(C, S, bps, seed, tonal=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(S)
    full = (1 << (bps - 1)) - 1
    x = np.stack([0.4 * np.sin(2 * np.pi * (220 + 37 * c) * t / 44100 + c) + 0.2 * np.sin(2 * np.pi * 3001 * t / 44100) for c in range(C)])
    x = x + 0.01 * rng.standard_normal((C, S)) if tonal else rng.uniform(-1, 1, (C, S))
    return np.clip(np.rint(x * full), -full - 1, full).astype(np.int64)


def check(pack, x, sr, bps, **kw):
    from egregora_amd import flacio
    buf = encode(x, sr, bps=bps, **kw)
    y, got_sr = flacio.read_flac_bytes(buf)
    want = (x.astype(np.float64) / float(1 << (bps - 1))).astype(np.float32).T
    assert got_sr == sr
    if x.shape[0] == 1:
        assert y.ndim == 1 and np.array_equal(y, want[:, 0])
    else:
        assert y.shape == want.shape and np.array_equal(y, want)
    return buf


@pytest.mark.parametrize("stereo", ["independent", "left_side", "side_right", "mid_side"])
@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
def test_fixed_predictors_and_stereo_modes(pack, stereo, order):
    x = pcm(2, 3000, 16, seed=order)
    check(pack, x, 44100, 16, blocksize=1152, stereo=stereo, plan=lambda fi, c: dict(kind=("fixed", order), porder=(fi + c) % 4))


def test_constant_verbatim_lpc_escape_rice2_wasted_bits(pack):
    x = pcm(2, 5000, 16, seed=7)
    x[1, 1152:2304] = -1234                         # a constant block
    x[0, 2304:3456] &= ~7                           # three wasted bits
    lpc = ([29, -14, 3], 6, 4)                      # quantised coefficients, precision 6, shift 4

    def plan(fi, c):
        if fi == 0:
            return dict(kind="verbatim")
        if fi == 1:
            return dict(kind="const") if c == 1 else dict(kind="lpc", lpc=lpc, porder=2)
        if fi == 2:
            return dict(kind=("fixed", 2), wasted=3, porder=1) if c == 0 else dict(kind="lpc", lpc=lpc, rice2=True, porder=3)
        return dict(kind=("fixed", 1), escape_first=True, porder=1) if c == 0 else dict(kind="lpc", lpc=([1], 2, 0), porder=0)
    check(pack, x, 48000, 16, blocksize=1152, plan=plan)


@pytest.mark.parametrize("bps,C,bs", [(8, 1, 192), (16, 1, 4096), (24, 2, 1000), (20, 3, 300), (12, 2, 70000 // 64)])
def test_widths_channels_and_block_size_fields(pack, bps, C, bs):
    x = pcm(C, 2 * bs + 17, bps, seed=bps, tonal=False)        # the last frame is short: explicit 8 / 16-bit block size field
    check(pack, x, 96000 if bps == 24 else 16000, bps, blocksize=bs, plan=lambda fi, c: dict(kind=("fixed", 1 + (fi + c) % 3), porder=0),
          max_frame_known=(bps != 20))


def test_corruption_is_loud(pack):
    from egregora_amd import flacio
    buf = bytearray(check(pack, pcm(1, 1500, 16, seed=3), 44100, 16))
    bad = bytearray(buf); bad[-40] ^= 0x10
    with pytest.raises(RuntimeError, match="CRC|sync|end of data|reserved|invalid"):
        flacio.read_flac_bytes(bytes(bad))
    with pytest.raises(RuntimeError, match="fLaC"):
        flacio.read_flac_bytes(b"RIFF....WAVE")
    assert flacio.crc8(b"123456789") == 0xF4 and flacio.crc16(b"123456789") == 0xFEE8      # CRC-8 / CRC-16-BUYPASS check values


def test_audio_io_sniffs_the_container(pack, tmp_path):
    from egregora_amd import audio_io, wavio
    x = pcm(2, 2000, 16, seed=11)
    (tmp_path / "a.flac").write_bytes(encode(x, 44100, plan=lambda fi, c: dict(kind=("fixed", 2), porder=2), stereo="mid_side"))
    wavio.write_wav_pcm16(str(tmp_path / "a.wav"), (x.T / 32768.0).astype(np.float32) * (32768.0 / 32767.0), 44100)
    yf, sf_ = audio_io.read_audio(str(tmp_path / "a.flac"))
    yw, sw = audio_io.read_audio(str(tmp_path / "a.wav"))
    assert sf_ == sw == 44100 and yf.shape == yw.shape == (2000, 2)
    assert np.abs(yf - yw).max() <= 1.0 / 32768.0
    (tmp_path / "a.ogg").write_bytes(b"OggS" + bytes(100))
    with pytest.raises(RuntimeError, match="WAV|FLAC"):
        audio_io.read_audio(str(tmp_path / "a.ogg"))
