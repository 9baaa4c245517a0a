"""Numpy model of the HIP Fat-Llama engine's index math (csrc/egr_fatllama.hip): Stockham stages with
the same (j,k,expand) addressing, the four-step split M = M1*M2 with two-table twiddles, the in-place
transposed-spectrum layout, and the row-pair real split / threshold / unsplit.  It is vectorised over
butterflies, uses float64, and exists so the addressing can be validated against numpy.fft on CPU
(tests/test_kernel_model.py) before/alongside the GPU parity tests.  Not product code.
"""
import numpy as np


def radix_schedule(L, allowed=(4, 2, 3, 5, 7, 11, 13)):
    """Same rule as egr::make_schedule (default build, EGR_RADIX_8_9): 8s, 4s, one 2, 9s, then odd primes ascending."""
    rad, n = [], L
    while n % 8 == 0:
        rad.append(8); n //= 8
    while n % 4 == 0:
        rad.append(4); n //= 4
    if n % 2 == 0:
        rad.append(2); n //= 2
    while n % 9 == 0:
        rad.append(9); n //= 9
    for p in (3, 5, 7, 11, 13):
        while n % p == 0:
            rad.append(p); n //= p
    if n != 1:
        return None
    return rad


def stockham_fft(x, inverse=False):
    """x: [..., L] complex. Forward DFT via the kernel's stage addressing. Inverse via the swap trick."""
    L = x.shape[-1]
    rad = radix_schedule(L)
    assert rad is not None, L
    a = x.astype(np.complex128)
    if inverse:
        a = a.imag + 1j * a.real
    tw = np.exp(-2j * np.pi * np.arange(L) / L)
    Ns = 1
    for R in rad:
        nb = L // R
        j = np.arange(nb)
        k = j % Ns
        out = np.empty_like(a)
        v = [a[..., j + t * nb] for t in range(R)]
        twstep = L // (Ns * R)
        v = [v[t] * tw[(k * t * twstep) % L] for t in range(R)]
        o = (j - k) * R + k
        WR = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
        for q in range(R):
            acc = 0
            for t in range(R):
                acc = acc + WR[q, t] * v[t]
            out[..., o + q * Ns] = acc
        a = out
        Ns *= R
    if inverse:
        a = a.imag + 1j * a.real
    return a


class Plan:
    def __init__(self, N, M1, M2):
        assert N % 2 == 0 and (N // 2) == M1 * M2
        self.N, self.M, self.M1, self.M2 = N, N // 2, M1, M2
        M = self.M
        self.T1 = np.exp(-2j * np.pi * np.arange(M1) / M1)          # W_M1^q  (= W_M^(q*M2))
        self.T2 = np.exp(-2j * np.pi * np.arange(M2) / M)           # W_M^s
        self.T3 = np.exp(-2j * np.pi * np.arange(M1) / N)           # W_N^k1
        self.T4 = np.exp(-2j * np.pi * np.arange(M2) / (2 * M2))    # W_N^(M1*k2)

    def tw_big(self, n2, k1):
        r = n2 * k1
        return self.T1[r // self.M2] * self.T2[r % self.M2]


def col_first(p, y, thr):
    """y real [N] -> A[k1][n2]  (threshold in time, FFT over n1, twiddle)."""
    d0 = np.where(np.abs(y) > thr, y, 0.0)
    z = (d0[0::2] + 1j * d0[1::2]).reshape(p.M1, p.M2)
    A = stockham_fft(z.T).T                                          # FFT along n1 for each column
    k1 = np.arange(p.M1)[:, None]
    n2 = np.arange(p.M2)[None, :]
    return A * p.tw_big(n2, k1)


def row_fused(p, A, thr):
    """A[k1][n2] -> B[k1][n2'] : FFT rows, real split + threshold + unsplit on (k, M-k) pairs, IFFT rows."""
    M1, M2, M = p.M1, p.M2, p.M
    Z = stockham_fft(A)                                              # Z[k1][k2] = Z[k1 + M1*k2]
    Zo = np.empty_like(Z)
    for ka in range(M1 // 2 + 1):
        kb = (M1 - ka) % M1
        if ka > kb:
            continue
        if ka != kb:
            k2 = np.arange(M2)
            pa, pb = (ka, k2), (kb, M2 - 1 - k2)
            skip_b = np.zeros(M2, bool)
        elif ka == 0:
            k2 = np.arange(M2 // 2 + 1)
            pa, pb = (0, k2), (0, (M2 - k2) % M2)
            skip_b = (k2 == (M2 - k2) % M2)
        else:  # ka == M1/2
            k2 = np.arange((M2 + 1) // 2)
            pa, pb = (ka, k2), (ka, M2 - 1 - k2)
            skip_b = (k2 == M2 - 1 - k2)
        Za, Zb = Z[pa], Z[pb]
        W = p.T3[ka] * p.T4[k2]
        E = 0.5 * (Za + np.conj(Zb))
        O = -0.5j * (Za - np.conj(Zb))
        WO = W * O
        Xk, Xmc = E + WO, E - WO
        Xk = np.where(np.abs(Xk) > thr, Xk, 0)
        Xmc = np.where(np.abs(Xmc) > thr, Xmc, 0)
        E2 = 0.5 * (Xk + Xmc)
        O2 = np.conj(W) * (0.5 * (Xk - Xmc))
        Za2 = (E2 + 1j * O2) / M
        Zb2 = np.conj(E2 - 1j * O2) / M
        Zo[pa] = Za2
        idx_b = (np.broadcast_to(pb[0], k2.shape)[~skip_b], pb[1][~skip_b])
        Zo[idx_b] = Zb2[~skip_b]
    return stockham_fft(Zo, inverse=True)                            # unnormalised IFFT over k2


def col_inverse(p, B):
    """B[k1][n2] -> time-domain packed z[n1][n2] (real d interleaved)."""
    k1 = np.arange(p.M1)[:, None]
    n2 = np.arange(p.M2)[None, :]
    t = B * np.conj(p.tw_big(n2, k1))
    z = stockham_fft(t.T, inverse=True).T
    d = np.empty(p.N)
    d[0::2] = z.real.reshape(-1)
    d[1::2] = z.imag.reshape(-1)
    return d


def col_mid(p, B):
    d = col_inverse(p, B)
    z = (d[0::2] + 1j * d[1::2]).reshape(p.M1, p.M2)
    A = stockham_fft(z.T).T
    k1 = np.arange(p.M1)[:, None]
    n2 = np.arange(p.M2)[None, :]
    return A * p.tw_big(n2, k1)


def ist(p, y, max_iter, thr):
    if max_iter == 0:
        return np.where(np.abs(y) > thr, y, 0.0)
    A = col_first(p, y, thr)
    for it in range(max_iter):
        B = row_fused(p, A, thr)
        if it + 1 < max_iter:
            A = col_mid(p, B)
    return col_inverse(p, B)


# ------------------------------------------------------------------------------------------------------------
# Three-level decomposition M = M1*M2*M3 (state [M1][M2][M3]) used for long inputs:
#   pass A : FFT over n1 (stride M2*M3), twiddle W_M^(c*k1), c = n2*M3+n3         (same kernel as 2-level k_col)
#   pass B : per k1 plane, FFT over n2 (stride M3), twiddle W_(M2*M3)^(n3*k2)      (k_col, forward-only / inverse-only)
#   pass C : rows of length M3; frequency k = o + R*k3 with o = k1 + M1*k2, R = M1*M2, stored at row
#            rho(o) = k1*M2 + k2; the real-split partner of (o, k3) is (R-o, M3-1-k3), or (0, (M3-k3)%M3) for o = 0.
# ------------------------------------------------------------------------------------------------------------
class Plan3:
    def __init__(self, N, M1, M2, M3):
        assert N % 2 == 0 and N // 2 == M1 * M2 * M3
        self.N, self.M, self.M1, self.M2, self.M3 = N, N // 2, M1, M2, M3
        self.R = M1 * M2

    def twA(self, c, k1):            # W_M^(c*k1)
        return np.exp(-2j * np.pi * ((c * k1) % self.M) / self.M)

    def twB(self, n3, k2):           # W_(M2*M3)^(n3*k2)
        Mp = self.M2 * self.M3
        return np.exp(-2j * np.pi * ((n3 * k2) % Mp) / Mp)

    def rho(self, o):
        return (o % self.M1) * self.M2 + o // self.M1


def _passA(p, S, inverse):
    """S [M1][M2*M3]: (inverse: x conj tw, IFFT over k1) or (forward: FFT over n1, x tw)."""
    k1 = np.arange(p.M1)[:, None]
    c = np.arange(p.M2 * p.M3)[None, :]
    if inverse:
        return stockham_fft((S * np.conj(p.twA(c, k1))).T, inverse=True).T
    return stockham_fft(S.T).T * p.twA(c, k1)


def _passB(p, S, inverse):
    """S [M1][M2][M3] planes."""
    k2 = np.arange(p.M2)[None, :, None]
    n3 = np.arange(p.M3)[None, None, :]
    S = S.reshape(p.M1, p.M2, p.M3)
    if inverse:
        t = S * np.conj(p.twB(n3, k2))
        return stockham_fft(np.swapaxes(t, 1, 2), inverse=True).swapaxes(1, 2).reshape(p.M1, -1)
    t = stockham_fft(np.swapaxes(S, 1, 2)).swapaxes(1, 2)
    return (t * p.twB(n3, k2)).reshape(p.M1, -1)


def _rows3(p, S, thr):
    M3, R, M, N = p.M3, p.R, p.M, p.N
    Z = stockham_fft(S.reshape(R, M3))                # row rho holds Z[o + R*k3]
    Zo = np.empty_like(Z)
    for o in range(R // 2 + 1):
        ob = (R - o) % R
        if o > ob:
            continue
        ra, rb = p.rho(o), p.rho(ob)
        if o != ob:
            k3 = np.arange(M3); pb = M3 - 1 - k3; skip = np.zeros(M3, bool)
        elif o == 0:
            k3 = np.arange(M3 // 2 + 1); pb = (M3 - k3) % M3; skip = (k3 == pb)
        else:
            k3 = np.arange((M3 + 1) // 2); pb = M3 - 1 - k3; skip = (k3 == pb)
        Za, Zb = Z[ra, k3], Z[rb, pb]
        W = np.exp(-2j * np.pi * (o + R * k3) / N)
        E = 0.5 * (Za + np.conj(Zb)); O = -0.5j * (Za - np.conj(Zb)); WO = W * O
        Xk, Xm = E + WO, E - WO
        Xk = np.where(np.abs(Xk) > thr, Xk, 0); Xm = np.where(np.abs(Xm) > thr, Xm, 0)
        E2 = 0.5 * (Xk + Xm); O2 = np.conj(W) * (0.5 * (Xk - Xm))
        Zo[ra, k3] = (E2 + 1j * O2) / M
        Zo[rb, pb[~skip]] = (np.conj(E2 - 1j * O2) / M)[~skip]
    return stockham_fft(Zo, inverse=True).reshape(p.M1, -1)


def ist3(p, y, max_iter, thr):
    d0 = np.where(np.abs(y) > thr, y, 0.0)
    if max_iter == 0:
        return d0
    S = (d0[0::2] + 1j * d0[1::2]).reshape(p.M1, -1)
    S = _passB(p, _passA(p, S, False), False)
    for it in range(max_iter):
        S = _passB(p, _rows3(p, S, thr), True)
        if it + 1 < max_iter:
            S = _passB(p, _passA(p, _passA(p, S, True), False), False)
    z = _passA(p, S, True)
    d = np.empty(p.N)
    d[0::2] = z.real.reshape(-1)
    d[1::2] = z.imag.reshape(-1)
    return d


# ---------------------------------------------------------------------------------------------------------------------------
# Lane / register maps of the two-barrier loop kernels (csrc/egr_fatllama_wl.h), element by element as the threads hold them.
def _W(T, e):
    return np.exp(-2j * np.pi * (np.asarray(e) % T) / T)


def wl_row_forward(x, n1=16, q=12):
    """k_row_wl<N1, Q>, forward half: row of L = N1 Q^2 -> regs[k1][lane c][d] = X[k1 + N1 (c + Q d)] (cross step: one thread per n2,
    radix N1 over n1 + twiddle W_L^(n2 k1); local step: Q lanes per block, radix Q over a, twiddle W_(Q^2)^(b c), Q x Q transpose,
    radix Q over b)."""
    qq = q * q
    L = n1 * qq
    F1 = _W(n1, np.outer(np.arange(n1), np.arange(n1)))
    Fq = _W(q, np.outer(np.arange(q), np.arange(q)))
    Y = (F1 @ x.reshape(n1, qq)) * _W(L, np.outer(np.arange(n1), np.arange(qq)))          # [k1][n2]
    regs = np.zeros((n1, q, q), complex)
    for k1 in range(n1):
        blk = Y[k1].reshape(q, q)                          # [a][b]
        Z = (Fq @ blk) * _W(qq, np.outer(np.arange(q), np.arange(q)))      # [c][b]
        regs[k1] = Z @ Fq                                   # [c][d] = sum_b Z[c][b] W_Q^(b d)
    return regs


def wl_row_inverse(regs):
    """the same steps backwards with conjugate twiddles (unnormalised)."""
    n1, q, _ = regs.shape
    qq = q * q
    L = n1 * qq
    F1 = np.conj(_W(n1, np.outer(np.arange(n1), np.arange(n1))))
    Fq = np.conj(_W(q, np.outer(np.arange(q), np.arange(q))))
    Y = np.zeros((n1, qq), complex)
    for k1 in range(n1):
        Z = (regs[k1] @ Fq) * np.conj(_W(qq, np.outer(np.arange(q), np.arange(q))))      # [c][b]
        Y[k1] = (Fq @ Z).reshape(qq)                         # [a][b] -> n2 = Q a + b
    return (F1 @ (Y * np.conj(_W(L, np.outer(np.arange(n1), np.arange(qq)))))).reshape(L)


def wl_row_partner(k1, c, d, n1=16, q=12):
    """(block, lane, register) of the real-split partner L - 1 - k of k = k1 + N1 (c + Q d): what the reversed-lane row-b unit holds."""
    return n1 - 1 - k1, q - 1 - c, q - 1 - d


def wl_col_mid(u):
    """k_col_wl on one column of 625 = 25 x 25 (without the four-step twiddles): inverse (thread b: radix 25 over a, x conj
    W_625^(b c), transpose, thread c: radix 25 over b) -> the thread holds t[c + 25 d], which is the input n = 25 a' + b' (b' = c) of
    its forward radix 25 -> x W_625^(b' c''), written into the row it read, transposed read, radix 25 over b'."""
    F = _W(25, np.outer(np.arange(25), np.arange(25)))
    T = _W(625, np.outer(np.arange(25), np.arange(25)))
    Z = (np.conj(F) @ u.reshape(25, 25)) * np.conj(T)        # [c][b]
    t_regs = Z @ np.conj(F)                                  # thread c, register d: t[c + 25 d]
    t = t_regs.T.reshape(625)                                # natural order n = c + 25 d
    Z2 = (F @ t_regs.T) * T                                  # thread b' = c holds a' = d: Z2[c''][b']
    X_regs = Z2 @ F                                          # thread c'', register d'': X[c'' + 25 d'']
    return t, X_regs.T.reshape(625)


def wl_inner(x, la, lb, forward=True):
    """k_colb_wl<LA, LB>: n2 = LB a + b -> k2 = c + LA d."""
    L = la * lb
    s = 1 if forward else -1
    Fa = _W(la, s * np.outer(np.arange(la), np.arange(la)))
    Fb = _W(lb, s * np.outer(np.arange(lb), np.arange(lb)))
    Z = (Fa @ x.reshape(la, lb)) * _W(L, s * np.outer(np.arange(la), np.arange(lb)))          # [c][b]
    return (Z @ Fb).T.reshape(L)                             # [c][d] -> k2 = c + LA d
