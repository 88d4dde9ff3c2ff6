"""Oracle (test infrastructure): a restatement of librosa's recursive constant-Q algorithm, used ONLY to
quantify how far the build's CQT specification (oracle/cqt.py: direct transform, zero padding) is from what
``librosa.cqt`` -- the call behind audio_complete.slice_C, util_audio.py:424-426 -- computes.

librosa (0.6 / 0.7, the versions the reference's API usage implies) is not installed and not vendored, so
this follows its published algorithm (Schoerkhuber & Klapuri 2010 as implemented in librosa/core/constantq.py)
from memory:
  * Q = filter_scale / (2^(1/bpo) - 1); the top octave's filters are complex sinusoids of length
    N_k = Q sr / f_k under a Hann window, L1-normalised, centred in n_fft = next power of two;
  * their FFT rows are sparsified (the smallest-magnitude entries carrying 1 % of each row's L1 mass are
    zeroed) and applied to an STFT of the signal with a rectangular window, hop H, reflect padding;
  * the signal is then low-pass filtered and decimated by two, the hop halved, the same basis applied for
    the next octave down, and so on; decimation keeps the signal's scale with a factor sqrt(2) per octave;
  * scale=True divides bin k by sqrt(N_k).
What cannot be reproduced is librosa's resampler (resampy 'kaiser_fast' / 'kaiser_best' filter tables);
scipy.signal.resample_poly's Kaiser-windowed FIR stands in for it.  The numbers this file produces are
therefore an ESTIMATE of librosa's output, good enough to bound the specification gap, not a pin.
"""
import numpy as np
from scipy import signal

from .audio import hann_periodic


def _sparsify_rows(M, quantile=0.01):
    out = M.copy()
    for r in range(M.shape[0]):
        mag = np.abs(M[r])
        order = np.argsort(mag)
        csum = np.cumsum(mag[order])
        cut = np.searchsorted(csum, quantile * csum[-1])
        out[r, order[:cut]] = 0
    return out


def _octave_basis(sr, freqs, Q, sparsity=0.01):
    lengths = Q * sr / freqs
    n_fft = int(2 ** np.ceil(np.log2(lengths.max())))
    basis = np.zeros((len(freqs), n_fft), dtype=np.complex128)
    for i, (f, ln) in enumerate(zip(freqs, lengths)):
        ilen = int(ln)
        n = np.arange(-ilen // 2, ilen // 2)
        sig = np.exp(2j * np.pi * f * n / sr) * hann_periodic(len(n))
        sig /= np.abs(sig).sum()                                   # norm=1
        start = (n_fft - len(sig)) // 2                            # pad_center
        basis[i, start:start + len(sig)] = sig
    basis *= lengths[:, None] / float(n_fft)
    fb = np.fft.fft(basis, n=n_fft, axis=1)[:, :n_fft // 2 + 1]
    return _sparsify_rows(fb, sparsity), n_fft, lengths


def _response(y, n_fft, hop, fft_basis):
    yp = np.pad(y, n_fft // 2, mode='reflect') if len(y) > n_fft // 2 else np.pad(y, n_fft // 2)
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    D = np.fft.rfft(yp[idx], axis=0)                               # window = ones
    return fft_basis.conj().dot(D) if False else fft_basis.dot(D)


def cqt_mag(y, sr, hop, fmin, n_bins, bins_per_octave, filter_scale=2.0):
    """|librosa.cqt(y, sr, hop_length=hop, fmin=fmin, n_bins=n_bins, bins_per_octave=bpo, filter_scale=2)|
    (estimate): float64 [n_bins, 1 + len(y) // hop]."""
    y = np.asarray(y, np.float64)
    n_oct = int(np.ceil(n_bins / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    freqs_all = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    top = freqs_all[-bins_per_octave:] if n_bins >= bins_per_octave else freqs_all
    Q = filter_scale / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    fft_basis, n_fft, _ = _octave_basis(sr, top[-n_filters:], Q)
    my_y, my_sr, my_hop = y, float(sr), int(hop)
    n_frames = 1 + len(y) // hop
    resp = []
    for o in range(n_oct):
        if o > 0:
            if my_hop % 2:
                raise ValueError('hop_length must be divisible by 2^(n_octaves - 1)')
            my_y = signal.resample_poly(my_y, 1, 2) * np.sqrt(2.0)  # audio.resample(..., scale=True)
            my_sr /= 2.0
            my_hop //= 2
            fft_basis = fft_basis * np.sqrt(2.0)
        r = _response(my_y, n_fft, my_hop, fft_basis)
        resp.append(r[:, :n_frames] if r.shape[1] >= n_frames else np.pad(r, ((0, 0), (0, n_frames - r.shape[1]))))
    C = np.vstack(resp[::-1])[-n_bins:]                            # lowest octave first
    lengths = Q * sr / freqs_all
    return np.abs(C) / np.sqrt(lengths)[:, None]
