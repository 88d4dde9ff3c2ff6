#!/usr/bin/env python3
"""Would a frequency-domain form of the 4 x 16 convolutions keep f32-equivalent accuracy?

CPU experiment, numpy/scipy only.  One 20 x 516 x 32 -> 32 layer of the calibrated timing head on its real
input activations:  (a) float64 direct convolution (truth), (b) float32 direct (what the reference's TensorFlow and
the oracle's e_cpu bar stand for), (c) float32 FFT along the 516-column axis (scipy.fft real transforms of length
576, complex64), per-frequency channel mixing in complex64, inverse transform.  Prints the errors of (b) and (c)
against (a) relative to max |output| and their ratio (the parity bar of tests/test_gpu_rdcnn.py is 4 x e_f32).
"""
import os
import sys
import numpy as np
import scipy.fft as sf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
from amt_saga import heads                                       # noqa: E402
from amt_saga.hyperparams import Hyperparams                     # noqa: E402
from oracle import rdcnn as orc                                  # noqa: E402

p = Hyperparams(N=2048)
h = heads.timming_classifier(p, weight_seed=107)
w = h.weights
rng = np.random.default_rng(0)
x0 = (rng.random((2, 20, 516, 1)) ** 2).astype(np.float32)
# activations entering layer 3: run layers 1-2 of the oracle
a = x0
for i in (1, 2):
    a = orc.sigmoid(orc.batchnorm(orc.conv2d_same(a, w['t0/conv%d/kernel' % i], w['t0/conv%d/bias' % i]), w, 't0/bn%d' % i))
k = w['t0/conv3/kernel']                                         # [4, 16, 32, 32]
KH, KW, C, F = k.shape
pt, pl = (KH - 1) // 2, (KW - 1) // 2


def direct(a, k, dt):
    a = a.astype(dt); k = k.astype(dt)
    B, H, W, _ = a.shape
    ap = np.zeros((B, H + KH - 1, W + KW - 1, C), dt)
    ap[:, pt:pt + H, pl:pl + W] = a
    out = np.zeros((B, H, W, F), dt)
    for dy in range(KH):
        for dx in range(KW):
            out += ap[:, dy:dy + H, dx:dx + W] @ k[dy, dx]
    return out


def fftconv(a, k, n=576):
    a = a.astype(np.float32); k = k.astype(np.float32)
    B, H, W, _ = a.shape
    A = sf.rfft(a, n=n, axis=2)                                  # complex64 [B, H, n/2+1, C]
    # correlation along x with 'same' padding: y[x] = sum_dx a[x + dx - pl] k[dx]  ->  kernel reversed and shifted
    kk = np.zeros((KH, n, C, F), np.float32)
    for dx in range(KW):
        kk[:, (pl - dx) % n] = k[:, dx]
    K = sf.rfft(kk, axis=1).astype(np.complex64)                 # [KH, n/2+1, C, F]
    assert A.dtype == np.complex64
    Y = np.zeros((B, H, n // 2 + 1, F), np.complex64)
    Ap = np.zeros((B, H + KH - 1, n // 2 + 1, C), np.complex64)
    Ap[:, pt:pt + H] = A
    for dy in range(KH):
        Y += np.einsum('bhfc,fcd->bhfd', Ap[:, dy:dy + H], K[dy]).astype(np.complex64)
    return sf.irfft(Y, n=n, axis=2)[:, :, :W].astype(np.float32)


truth = direct(a, k, np.float64)
f32 = direct(a, k, np.float32)
ff = fftconv(a, k)
sc = np.abs(truth).max()
e32 = np.abs(f32 - truth).max() / sc
eff = np.abs(ff - truth).max() / sc
r32 = np.sqrt(np.mean((f32 - truth) ** 2)) / sc
rff = np.sqrt(np.mean((ff - truth) ** 2)) / sc
print('max |out| %.3f' % sc)
print('direct f32 : max err %.3e  rms %.3e   (relative to max |out|)' % (e32, r32))
print('FFT    f32 : max err %.3e  rms %.3e   ratio to direct f32: max %.2f  rms %.2f' % (eff, rff, eff / e32, rff / r32))
