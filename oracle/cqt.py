"""Oracle (test infrastructure): the build's constant-Q slice spec.

Stands where the reference calls ``np.abs(librosa.cqt(self.wf, sr, fmin=
note_to_hz(lowest_note), n_bins, bins_per_octave=12*bins_per_tone,
filter_scale=2, hop_length=self.hl))`` and keeps ``_resize(C[:, s:t], target)``
(/root/reference/util_audio.py:411-434).

PARITY UNPINNED vs librosa: librosa is absent and unversioned, its cqt is a
recursive multi-rate approximation (early downsampling, resampling filters,
sparsified frequency-domain kernels at 1 % threshold), and the reference
records no CQT value anywhere (SURVEY 7 hard part 1, 8c).  This file therefore
DEFINES the transform both the CPU baseline and the HIP kernel compute: the
direct (un-approximated) constant-Q transform that librosa's algorithm
approximates, with librosa's documented parameterisation:

    Q     = filter_scale / (2**(1/bpo) - 1),  filter_scale = 2   (:426 hard-codes 2)
    f_k   = fmin * 2**(k/bpo)
    N_k   = ceil(Q * sr / f_k)                 filter length in samples
    w_k   = periodic Hann(N_k), L1-normalised (norm=1);  sum(w_k) = N_k/2
    C[k,t]= sqrt(N_k) * | sum_n x[c_t - floor(N_k/2) + n] * w_k[n]/sum(w_k)
                                  * exp(-2 pi i f_k (.)/sr) |,   c_t = t*hop
            (scale=True multiplies by sqrt(N_k) after the length-normalised
             response; x is zero outside the window -- pad_mode differs from
             librosa's reflect, which cannot be reproduced for N_k > len(x))
    frequency is quantised to phase_inc_k = rint(f_k/sr * 2**32) (uint32 cycles
    per sample) so that CPU and GPU use bit-identical oscillator phases.

Only the (<= target) frames that survive ``_resize`` are evaluated.
"""
import numpy as np

FILTER_SCALE = 2.0


def note_to_midi(note):
    """librosa.note_to_midi for the spellings the reference uses ('A0','C8',
    'C4', util_audio.py:414; training.py passes midi_to_note(pitch))."""
    if isinstance(note, (int, np.integer)):
        return int(note)
    pitch_map = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}
    name = note[0].upper()
    rest = note[1:]
    acc = 0
    while rest and rest[0] in '#b':
        acc += 1 if rest[0] == '#' else -1
        rest = rest[1:]
    octave = int(rest) if rest else 0
    return 12 * (octave + 1) + pitch_map[name] + acc


def cqt_table(sr, fmin_hz, n_bins, bins_per_octave):
    """(phase_inc uint32[n_bins], length int32[n_bins], freq float64[n_bins])."""
    k = np.arange(n_bins, dtype=np.float64)
    freq = float(fmin_hz) * 2.0 ** (k / bins_per_octave)
    Q = FILTER_SCALE / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    length = np.ceil(Q * sr / freq).astype(np.int64)
    inc = np.rint(freq / sr * 2.0 ** 32).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    return inc.astype(np.uint32), length.astype(np.int32), freq


def cqt_frames(x, frames, phase_inc, length, hop, complex_out=False):
    """|C[k, t]| for the listed STFT-frame indices (-1 -> zero column).
    x float [L]; returns float64 [n_bins, len(frames)].

    complex_out (util_audio.py:428, magnitude_only=False: the complex CQT is handed on as librosa returns it): complex128
    C[k, t] with the phase referred to the frame's CENTRE, sample t * hop -- the analysis filter's own phase is zero there,
    the convention of a centred filter bank; in the integer oscillator below that is the factor
    exp(+i phi(t * hop)) on the absolute-phase sum.

    The oscillator phase of absolute sample m is (m * inc mod 2^32) / 2^32 turns.
    Because that map is additive mod 2^32, exp(-i phi(a+n)) = exp(-i phi(a)) *
    exp(-i phi(n)) exactly, and the unit factor exp(-i phi(a)) drops out of the
    magnitude -- so one basis per bin (relative index n) serves every frame."""
    x = np.asarray(x, dtype=np.float64)
    L = len(x)
    out = np.zeros((len(length), len(frames)), dtype=np.complex128 if complex_out else np.float64)
    for k in range(len(length)):
        nk = int(length[k])
        inc = int(phase_inc[k])
        n = np.arange(nk, dtype=np.uint64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(nk) / nk)
        ph = ((n * np.uint64(inc)) & np.uint64(0xFFFFFFFF)).astype(np.float64)
        basis = w * np.exp(-1j * ph * (2.0 * np.pi / 2.0 ** 32))
        for j, t in enumerate(frames):
            if t < 0:
                continue
            a = int(t) * hop - nk // 2
            lo, hi = max(a, 0), min(a + nk, L)
            if hi <= lo:
                continue
            s = np.dot(x[lo:hi], basis[lo - a:hi - a])
            if complex_out:
                # basis index n carries phi(n); absolute sample a + n carries phi(a) + phi(n): s e^{-i phi(a)} is the
                # absolute-phase sum, and e^{+i phi(t hop)} refers it to the frame centre -- together e^{+i phi(t hop - a)}
                rel = ((int(t) * hop - a) * inc) & 0xFFFFFFFF
                out[k, j] = s * np.exp(1j * rel * (2.0 * np.pi / 2.0 ** 32)) * 2.0 / np.sqrt(nk)
            else:
                out[k, j] = np.abs(s) * 2.0 / np.sqrt(nk)
    return out


def slice_C_frames(n_frames_total, s, t, target):
    """Source frame of every output column of _resize(C[:, s:t], target)
    (util_audio.py:431-434 with the tile/crop rule of :384-409); -1 = zeros."""
    from .audio import resize_index_map
    s = max(0, min(s, n_frames_total))
    t = max(s, min(t, n_frames_total))
    rel = resize_index_map(t - s, target)
    return np.where(rel < 0, -1, rel + s)


def cqt_window_max(x, phase_inc, length, hop):
    """max over every bin and EVERY frame t = 0 .. len(x)//hop of |C[k, t]| -- the song-level normalisers
    ``np.max(slice_C(0, duration, n_frames, ...))`` of training.py:271-282 for the build's CQT.
    Evaluated per bin in O(len(x)) with cumulative sums (float64): with w(n) = 1/2 - 1/2 cos(theta n),
    sum_n x[a+n] w(n) e^{-i phi (a+n)} = 1/2 S0 - 1/4 (e^{-i theta a} S+ + e^{+i theta a} S-) over the frame's
    support, S0 = sum z, S+- = sum z e^{+-i theta m}, z[m] = x[m] e^{-i phi m}; equal to cqt_frames() on all frames."""
    x = np.asarray(x, dtype=np.float64)
    L = len(x)
    T = 1 + L // hop
    m = np.arange(L, dtype=np.uint64)
    best = 0.0
    t = np.arange(T, dtype=np.int64)
    for k in range(len(length)):
        nk, inc = int(length[k]), int(phase_inc[k])
        ph = ((m * np.uint64(inc)) & np.uint64(0xFFFFFFFF)).astype(np.float64) * (2.0 * np.pi / 2.0 ** 32)
        z = x * np.exp(-1j * ph)
        th = 2.0 * np.pi / nk
        e = np.exp(1j * th * (np.arange(L) % nk))                  # e^{i theta m} (period nk: exact reduction)
        c0 = np.concatenate(([0], np.cumsum(z)))
        cp = np.concatenate(([0], np.cumsum(z * e)))
        cm = np.concatenate(([0], np.cumsum(z * np.conj(e))))
        a = t * hop - nk // 2
        lo, hi = np.clip(a, 0, L), np.clip(a + nk, 0, L)
        s0, sp, sm = c0[hi] - c0[lo], cp[hi] - cp[lo], cm[hi] - cm[lo]
        rot = np.exp(-1j * th * (a % nk))                          # e^{-i theta a}
        C = 0.5 * s0 - 0.25 * (rot * sp + np.conj(rot) * sm)
        best = max(best, float(np.abs(C).max() * 2.0 / np.sqrt(nk)))
    return best
