#!/usr/bin/env python3
"""CPU model (numpy, float64) of the FFT-domain form of one 4 x 16 convolution layer of the timing head, written to
pin down the algebra the HIP kernels implement (amt_fftconv.hip) before any of them existed:

  * rows of W = 516 positions are transformed with NF = 576-point COMPLEX FFTs of channel PAIRS, z_p = a_{2p} + i a_{2p+1}
    (no real-FFT separation pass);
  * per frequency pair (f, NF - f) the layer is one real GEMM with K = 4 row taps x [Z_p[f], Z_p[NF-f]] (256 reals)
    and N = [W_q[f], W_q[NF-f]] (64 reals): the (de)interleaving of the packed channels lives in the transformed
    kernel matrices;
  * the inverse FFT of W_q = Y_{2q} + i Y_{2q+1} returns two real output channels at once.

Run: python scripts/fftconv_model.py   (asserts the model against a direct float64 convolution)."""
import numpy as np

NF = 576


def direct(a, k):
    """a [H][W][Ci], k [KH][KW][Ci][Co], Keras 'same' (pad_before = (k-1)//2)."""
    H, W, Ci = a.shape
    KH, KW, _, Co = k.shape
    pt, pl = (KH - 1) // 2, (KW - 1) // 2
    ap = np.zeros((H + KH - 1, W + KW - 1, Ci))
    ap[pt:pt + H, pl:pl + W] = a
    y = np.zeros((H, W, Co))
    for dy in range(KH):
        for dx in range(KW):
            y += ap[dy:dy + H, dx:dx + W] @ k[dy, dx]
    return y


def kernel_spectrum(k):
    """Kf[dy][f][ci][co] = sum_dx K[dy][dx][ci][co] e^{+2 pi i f (dx - pl) / NF}: Y[w] = IDFT(Af Kf)[w]."""
    KH, KW, Ci, Co = k.shape
    pl = (KW - 1) // 2
    f = np.arange(NF)[:, None]
    ph = np.exp(2j * np.pi * f * (np.arange(KW)[None, :] - pl) / NF)          # [f][dx]
    return np.einsum('fd,ydio->yfio', ph, k)


def pair_matrices(kf):
    """Real GEMM matrices G[fp][K = (dy, side s in {f, NF-f}, p, re/im)][N = (side t, q, re/im)] for fp = 0 .. NF/2."""
    KH, _, Ci, Co = kf.shape
    P, Q = Ci // 2, Co // 2
    G = np.zeros((NF // 2 + 1, KH * 2 * P * 2, 2 * Q * 2))
    for fp in range(NF // 2 + 1):
        for t, fo in enumerate((fp, (NF - fp) % NF)):            # output side: W_q[fo] = Y_{2q}[fo] + i Y_{2q+1}[fo]
            # Y_c[fo] = sum_dy sum_ci A_ci[fo] Kf[dy][fo][ci][c];  A_{2p}[fo] = (Z_p[fo] + conj Z_p[NF-fo]) / 2,
            # A_{2p+1}[fo] = (Z_p[fo] - conj Z_p[NF-fo]) / (2i)
            # -> coefficient of Z_p[fo]:          (Kf[2p] - i Kf[2p+1]) / 2
            #    coefficient of conj Z_p[NF-fo]:  (Kf[2p] + i Kf[2p+1]) / 2
            for dy in range(KH):
                ke, ko = kf[dy, fo, 0::2, :], kf[dy, fo, 1::2, :]               # [p][c]
                cz = (ke - 1j * ko) / 2                                          # [p][c] on Z_p[fo]
                cc = (ke + 1j * ko) / 2                                          # on conj Z_p[NF-fo]
                wz = cz[:, 0::2] + 1j * cz[:, 1::2]                              # -> W_q: [p][q]
                wc = cc[:, 0::2] + 1j * cc[:, 1::2]
                # which input side holds Z[fo] / Z[NF-fo]?  side 0 = Z[fp], side 1 = Z[NF-fp]
                s_z, s_c = (0, 1) if t == 0 else (1, 0)
                for p in range(P):
                    for q in range(Q):
                        kz = ((dy * 2 + s_z) * P + p) * 2
                        kc = ((dy * 2 + s_c) * P + p) * 2
                        n = (t * Q + q) * 2
                        # (zr + i zi) * (wr + i wi): re += zr wr - zi wi, im += zr wi + zi wr
                        G[fp, kz, n] += wz[p, q].real; G[fp, kz + 1, n] += -wz[p, q].imag
                        G[fp, kz, n + 1] += wz[p, q].imag; G[fp, kz + 1, n + 1] += wz[p, q].real
                        # conj(z) * w: (zr - i zi)(wr + i wi): re += zr wr + zi wi, im += zr wi - zi wr
                        G[fp, kc, n] += wc[p, q].real; G[fp, kc + 1, n] += wc[p, q].imag
                        G[fp, kc, n + 1] += wc[p, q].imag; G[fp, kc + 1, n + 1] += -wc[p, q].real
        if fp == (NF - fp) % NF:
            # self-paired frequencies (0 and NF/2): both input sides are the same bin and both output sides too; keep
            # side 0 only -- fold the side-1 input rows into side 0 and drop the side-1 outputs
            for dy in range(KH):
                for p in range(P):
                    for e in range(2):
                        k0 = ((dy * 2 + 0) * P + p) * 2 + e
                        k1 = ((dy * 2 + 1) * P + p) * 2 + e
                        G[fp, k0, :] += G[fp, k1, :]
                        G[fp, k1, :] = 0
            G[fp, :, 2 * Q:] = 0
    return G


def fftconv(a, k):
    H, W, Ci = a.shape
    KH, KW, _, Co = k.shape
    P, Q = Ci // 2, Co // 2
    pt = (KH - 1) // 2
    z = np.zeros((H, NF, P), complex)
    z[:, :W] = a[:, :, 0::2] + 1j * a[:, :, 1::2]
    Z = np.fft.fft(z, axis=1)                                                   # [h][f][p]
    G = pair_matrices(kernel_spectrum(k))
    Wq = np.zeros((H, NF, Q), complex)
    for fp in range(NF // 2 + 1):
        fm = (NF - fp) % NF
        # A rows [h][K]: taps dy read row h + dy - pt (zero outside)
        A = np.zeros((H, KH, 2, P, 2))
        for dy in range(KH):
            for h in range(H):
                hs = h + dy - pt
                if 0 <= hs < H:
                    A[h, dy, 0, :, 0], A[h, dy, 0, :, 1] = Z[hs, fp].real, Z[hs, fp].imag
                    A[h, dy, 1, :, 0], A[h, dy, 1, :, 1] = Z[hs, fm].real, Z[hs, fm].imag
        out = A.reshape(H, -1) @ G[fp]                                           # [h][(t, q, re/im)]
        out = out.reshape(H, 2, Q, 2)
        Wq[:, fp] = out[:, 0, :, 0] + 1j * out[:, 0, :, 1]
        if fm != fp:
            Wq[:, fm] = out[:, 1, :, 0] + 1j * out[:, 1, :, 1]
    y = np.fft.ifft(Wq, axis=1)[:, :W]
    out = np.zeros((H, W, Co))
    out[:, :, 0::2], out[:, :, 1::2] = y.real, y.imag
    return out


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    a = rng.random((20, 516, 8))
    k = rng.standard_normal((4, 16, 8, 6)) * 0.1
    ref = direct(a, k)
    got = fftconv(a, k)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print('fft-domain vs direct: max rel err %.2e' % err)
    assert err < 1e-12
