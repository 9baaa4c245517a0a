"""Test-only FLAC ENCODER, written from the published format description (the build image has no flac / libsndfile to make
fixtures with).  It exists to exercise every branch of the product's decoder (egregora_amd.flacio): all subframe types, fixed
orders 0-4, LPC with quantised coefficients, Rice / Rice2 partitions with per-partition parameters and escape codes, wasted bits,
the four stereo modes, 8-bit / 16-bit block-size fields, CRC-8 / CRC-16.  Bit-exact round trips of integer PCM are the check."""
import numpy as np


def _crc(data, poly, width):
    top, mask, c = 1 << (width - 1), (1 << width) - 1, 0
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


class W:
    def __init__(self):
        self.bits = []

    def u(self, v, n):
        assert 0 <= v < (1 << n) or n == 0, (v, n)
        self.bits.append(format(v, "b").zfill(n) if n else "")

    def i(self, v, n):
        self.u(v & ((1 << n) - 1), n)

    def unary(self, q):
        self.bits.append("0" * q + "1")

    def bytes(self):
        s = "".join(self.bits)
        s += "0" * (-len(s) % 8)
        return int(s, 2).to_bytes(len(s) // 8, "big") if s else b""


def _utf8(n):
    if n < 0x80:
        return bytes([n])
    out, k = [], 0
    while n >= (0x40 >> k):
        out.append(0x80 | (n & 0x3F)); n >>= 6; k += 1
    lead = ((0xFF << (7 - k)) & 0xFF) | n
    return bytes([lead] + out[::-1])


def _rice(w, res, order, blocksize, porder, rice2, escape_first):
    w.u(1 if rice2 else 0, 2)
    w.u(porder, 4)
    pb = 5 if rice2 else 4
    n0, pos = blocksize >> porder, 0
    for part in range(1 << porder):
        n = n0 - (order if part == 0 else 0)
        seg = res[pos:pos + n]; pos += n
        if escape_first and part == 0:
            raw = max(1, max((int(v).bit_length() + 1 for v in seg), default=1))
            w.u((1 << pb) - 1, pb); w.u(raw, 5)
            for v in seg:
                w.i(int(v), raw)
            continue
        z = [(2 * int(v)) if v >= 0 else (-2 * int(v) - 1) for v in seg]
        mean = (sum(z) / len(z)) if z else 0
        k = min(max(int(np.log2(mean + 1)), 0), (1 << pb) - 2)
        w.u(k, pb)
        for u in z:
            w.unary(u >> k)
            w.u(u & ((1 << k) - 1), k)


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _residual_of(x, coefs, shift):
    order = len(coefs)
    return [int(x[n]) - (sum(c * int(x[n - 1 - i]) for i, c in enumerate(coefs)) >> shift) for n in range(order, len(x))]


def _subframe(w, x, bps, kind, porder=0, rice2=False, escape_first=False, wasted=0, lpc=None):
    """kind: 'const' | 'verbatim' | ('fixed', order) | 'lpc' (lpc = (coefs, precision, shift))."""
    x = [int(v) >> wasted for v in x]
    bps -= wasted
    w.u(0, 1)
    code = 0 if kind == "const" else 1 if kind == "verbatim" else (8 + kind[1]) if kind[0] == "fixed" else 31 + len(lpc[0])
    w.u(code, 6)
    if wasted:
        w.u(1, 1); w.unary(wasted - 1)
    else:
        w.u(0, 1)
    if kind == "const":
        w.i(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            w.i(v, bps)
    elif kind[0] == "fixed":
        o = kind[1]
        for v in x[:o]:
            w.i(v, bps)
        _rice(w, _residual_of(x, _FIXED[o], 0), o, len(x), porder, rice2, escape_first)
    else:
        coefs, prec, shift = lpc
        for v in x[:len(coefs)]:
            w.i(v, bps)
        w.u(prec - 1, 4); w.i(shift, 5)
        for c in coefs:
            w.i(c, prec)
        _rice(w, _residual_of(x, coefs, shift), len(coefs), len(x), porder, rice2, escape_first)


def encode(pcm, sr, bps=16, blocksize=1152, stereo="independent", plan=None, max_frame_known=True):
    """pcm: int array [C][S].  plan(frame_index, channel) -> dict of _subframe options (kind, porder, ...)."""
    pcm = np.asarray(pcm, dtype=np.int64)
    C, S = pcm.shape
    frames = []
    for fi, start in enumerate(range(0, S, blocksize)):
        blk = pcm[:, start:start + blocksize]
        n = blk.shape[1]
        w = W()
        w.u(0x3FFE, 14); w.u(0, 1); w.u(0, 1)
        std = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
        bs_code = std.get(n, 6 if n <= 256 else 7)
        w.u(bs_code, 4)
        w.u({44100: 9, 48000: 10, 16000: 5, 96000: 11}.get(sr, 0), 4)
        ca = {"independent": C - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[stereo]
        w.u(ca, 4)
        w.u({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0), 3); w.u(0, 1)
        for b in _utf8(fi):
            w.u(b, 8)
        if bs_code == 6:
            w.u(n - 1, 8)
        elif bs_code == 7:
            w.u(n - 1, 16)
        hdr = w.bytes()
        w.u(_crc(hdr, 0x07, 8), 8)
        chans, widths = [blk[c] for c in range(C)], [bps] * C
        if stereo == "left_side":
            chans, widths = [blk[0], blk[0] - blk[1]], [bps, bps + 1]
        elif stereo == "side_right":
            chans, widths = [blk[0] - blk[1], blk[1]], [bps + 1, bps]
        elif stereo == "mid_side":
            chans, widths = [(blk[0] + blk[1]) >> 1, blk[0] - blk[1]], [bps, bps + 1]
        for c, (x, wd) in enumerate(zip(chans, widths)):
            opt = dict(kind=("fixed", 2))
            if plan:
                opt = plan(fi, c)
            _subframe(w, x, wd, **opt)
        body = w.bytes()
        frames.append(body + _crc(body, 0x8005, 16).to_bytes(2, "big"))
    mx = max(len(f) for f in frames) if max_frame_known else 0
    info = (blocksize.to_bytes(2, "big") * 2 + (0).to_bytes(3, "big") + mx.to_bytes(3, "big") +
            ((sr << 44) | ((C - 1) << 41) | ((bps - 1) << 36) | S).to_bytes(8, "big") + bytes(16))
    return b"fLaC" + bytes([0x80]) + len(info).to_bytes(3, "big") + info + b"".join(frames)
