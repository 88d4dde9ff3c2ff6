#!/usr/bin/env python3
"""CPU model (numpy, float64) of the PACKED-IMAGE FFT-domain form proposed for the 10 x 64, 64 -> 64 (4 x 16) layers of the
timing nets -- written before any kernel, to pin down the geometry:

  * the whole image goes into ONE sequence: row h of the image occupies positions (h + PT) * P .. (h + PT) * P + W - 1 with
    a pitch P = W + KW - 1 = 79 (the gaps are zeros), PT = 1 empty row slot in front and KH - 1 - PT = 2 behind:
    (10 + 3) * 79 = 1027 <= NF = 1152 = 24 x 48;
  * the 2-D "same" convolution is then the 1-D circular convolution of that sequence with the kernel's taps at offsets
    (dy - PT) * P + (dx - PL): the four row taps live in the transformed kernel -- ONE matrix per frequency instead of a
    K = 4 x ... contraction over image rows -- and every window is one GEMM row;
  * channel pairs ride one complex transform, z_p = a_2p + i a_2p+1, exactly as in scripts/fftconv_model.py: per frequency
    pair (f, NF - f) a real GEMM with K = [Z_p[f], Z_p[NF-f]] x 32 pairs x (re, im) = 128 and N = 128;
  * a chain re-zeroes the gaps / pad slots after the epilogue (sigmoid(0 + t) != 0 there) and transforms forward again.

Run: python scripts/fftconv_packed_model.py   (asserts against a direct float64 convolution, single layer and a chain of two)."""
import numpy as np

H, W, KH, KW = 10, 64, 4, 16
PT, PL = (KH - 1) // 2, (KW - 1) // 2
P = W + KW - 1                    # 79
NF = 1152
assert (H + KH - 1) * P <= NF


def direct(a, k):
    ap = np.zeros((H + KH - 1, W + KW - 1, a.shape[2]))
    ap[PT:PT + H, PL:PL + W] = a
    y = np.zeros((H, W, k.shape[3]))
    for dy in range(KH):
        for dx in range(KW):
            y += ap[dy:dy + H, dx:dx + W] @ k[dy, dx]
    return y


def pack(a):
    """[H][W][C] -> [NF][C]: row h at (h + PT) * P."""
    s = np.zeros((NF, a.shape[2]))
    for h in range(H):
        s[(h + PT) * P:(h + PT) * P + W] = a[h]
    return s


def unpack(s):
    return np.stack([s[(h + PT) * P:(h + PT) * P + W] for h in range(H)])


def kernel_spectrum(k):
    """Kf[f][ci][co] = sum_{dy,dx} K[dy][dx][ci][co] e^{+2 pi i f ((dy - PT) P + (dx - PL)) / NF}:
    out[m] = sum_taps in[m + (dy - PT) P + (dx - PL)] K[dy][dx]  <=>  Out_f = In_f Kf[f]."""
    off = ((np.arange(KH)[:, None] - PT) * P + (np.arange(KW)[None, :] - PL)).reshape(-1)      # [taps]
    ph = np.exp(2j * np.pi * np.arange(NF)[:, None] * off[None, :] / NF)                      # [f][tap]
    return np.einsum('ft,tio->fio', ph, k.reshape(KH * KW, k.shape[2], k.shape[3]))


def pair_matrices(kf):
    """G[fp][K = (side s, pair p, re/im)][N = (side t, pair q, re/im)], fp = 0 .. NF/2 (fftconv_model.pair_matrices with one
    row tap)."""
    Ci, Co = kf.shape[1], kf.shape[2]
    Pn, Q = Ci // 2, Co // 2
    G = np.zeros((NF // 2 + 1, 2 * Pn * 2, 2 * Q * 2))
    for fp in range(NF // 2 + 1):
        for t, fo in enumerate((fp, (NF - fp) % NF)):
            ke, ko = kf[fo, 0::2, :], kf[fo, 1::2, :]
            cz, cc = (ke - 1j * ko) / 2, (ke + 1j * ko) / 2
            wz = cz[:, 0::2] + 1j * cz[:, 1::2]
            wc = cc[:, 0::2] + 1j * cc[:, 1::2]
            s_z, s_c = (0, 1) if t == 0 else (1, 0)
            for p in range(Pn):
                kz, kc = (s_z * Pn + p) * 2, (s_c * Pn + p) * 2
                n = (t * Q + np.arange(Q)) * 2
                G[fp, kz, n] += wz[p].real; G[fp, kz + 1, n] += -wz[p].imag
                G[fp, kz, n + 1] += wz[p].imag; G[fp, kz + 1, n + 1] += wz[p].real
                G[fp, kc, n] += wc[p].real; G[fp, kc + 1, n] += wc[p].imag
                G[fp, kc, n + 1] += wc[p].imag; G[fp, kc + 1, n + 1] += -wc[p].real
        if fp == (NF - fp) % NF:                       # self-paired bins: keep side 0 only
            half = Pn * 2
            G[fp, :half, :] += G[fp, half:, :]
            G[fp, half:, :] = 0
            G[fp, :, Q * 2:] = 0
    return G


def forward(a):
    """[H][W][C] -> pair spectra X[fp][(side, pair, re/im)] (one GEMM row per frequency pair)."""
    s = pack(a)
    z = np.fft.fft(s[:, 0::2] + 1j * s[:, 1::2], axis=0)             # [NF][pairs]
    Pn = z.shape[1]
    X = np.zeros((NF // 2 + 1, 2 * Pn * 2))
    for fp in range(NF // 2 + 1):
        for sd, f in enumerate((fp, (NF - fp) % NF)):
            X[fp, (sd * Pn + np.arange(Pn)) * 2] = z[f].real
            X[fp, (sd * Pn + np.arange(Pn)) * 2 + 1] = z[f].imag
    return X


def inverse(Y, Q):
    """pair spectra [fp][(side, q, re/im)] -> [H][W][2 Q] (real channels 2q, 2q + 1 = re, im of the inverse transform)."""
    w = np.zeros((NF, Q), complex)
    for fp in range(NF // 2 + 1):
        for sd, f in enumerate((fp, (NF - fp) % NF)):
            if sd == 1 and f == fp:
                continue
            w[f] = Y[fp, (sd * Q + np.arange(Q)) * 2] + 1j * Y[fp, (sd * Q + np.arange(Q)) * 2 + 1]
    y = np.fft.ifft(w, axis=0)
    s = np.zeros((NF, 2 * Q))
    s[:, 0::2], s[:, 1::2] = y.real, y.imag
    return unpack(s)


def layer(a, G, Q):
    X = forward(a)
    Y = np.einsum('fk,fkn->fn', X, G)
    return inverse(Y, Q)


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    C = 64
    a = rng.standard_normal((H, W, C))
    k1 = rng.standard_normal((KH, KW, C, C)) / np.sqrt(KH * KW * C)
    k2 = rng.standard_normal((KH, KW, C, C)) / np.sqrt(KH * KW * C)
    G1, G2 = pair_matrices(kernel_spectrum(k1)), pair_matrices(kernel_spectrum(k2))
    y1 = layer(a, G1, C // 2)
    ref1 = direct(a, k1)
    e1 = np.abs(y1 - ref1).max() / np.abs(ref1).max()
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    y2 = layer(sig(y1), G2, C // 2)                   # the chain: epilogue on the valid positions, gaps re-zeroed by pack()
    ref2 = direct(sig(ref1), k2)
    e2 = np.abs(y2 - ref2).max() / np.abs(ref2).max()
    print('packed-image FFT conv vs direct: layer %.2e, chain of two %.2e; GEMM K = %d, N = %d per frequency pair, %d pairs; '
          '%.0f KB per frequency tensor and window' % (e1, e2, G1.shape[1], G1.shape[2], G1.shape[0], G1.shape[0] * G1.shape[1] * 4 / 1024))
    assert e1 < 1e-12 and e2 < 1e-12
