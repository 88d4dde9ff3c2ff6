"""audio_complete (reference call surface) and AudioBatch (B windows at once).

Mirrors /root/reference/util_audio.py:32-527.  Spectra live in HBM frame-major
([T][ldf], ldf = F rounded up to 4) and are transposed to the reference's
[F, T] numpy layout only at the property boundary.  Every numeric step runs in
the HIP library (include/amt_saga.h); numpy is used for the host-visible views
and the integer index maps (_resize tables, frame<->second maps).
"""
import bisect
import ctypes as C
import math

import os

import numpy as np
import torch

from . import _lib
from .device import empty, ptr, require_gpu, stream_ptr, to_dev, zeros

_PLANS = {}


def _stride0(t):
    """Elements between consecutive items of dim 0 (size-1 dims may carry stride 0)."""
    return t.stride(0) if t.shape[0] > 1 else max(t[0].numel(), 1)


def _plan(n_fft, hop, center):
    key = (int(n_fft), int(hop), bool(center), torch.cuda.current_device())
    if key not in _PLANS:
        lib = _lib.load()
        require_gpu()
        h = C.c_void_p()
        _lib.check(lib.amt_stft_plan_create(C.byref(h), int(n_fft), int(hop), int(bool(center))))
        _PLANS[key] = h
    return _PLANS[key]


# Row pitch of the spectrograms in floats.  AMT_LDF_ALIGN=32 (128-byte rows) is a measured experiment, not a supported
# mode: the STFT's write traffic drops from 2.39 to 2.25 MB per window (3.55 -> 3.41 in total, PMC) at unchanged speed, the
# other kernels carry 2.7 % more bytes, and the host-side window-management helpers (gather_frames) assume the default
_LDF_ALIGN = int(os.environ.get('AMT_LDF_ALIGN', '4'))


def ldf_of(n_fft):
    return ((n_fft // 2 + 1) + _LDF_ALIGN - 1) // _LDF_ALIGN * _LDF_ALIGN


def fft_frequencies(sr, n_fft):
    return np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)


def midi_to_hz(m):
    return 440.0 * (2.0 ** ((np.asanyarray(m, dtype=np.float64) - 69.0) / 12.0))


def resize_source_frames(t, target):
    """Index form of audio_complete._resize (util_audio.py:384-409): for a
    source of t frames, the source frame of each of `target` output columns;
    -1 = zero column (t == 0)."""
    if t <= 0:
        return np.full(target, -1, dtype=np.int32)
    if t == target:
        return np.arange(target, dtype=np.int32)
    if t < 3:
        return np.array([0] + [t - 1] * (target - 1), dtype=np.int32)
    if t < target:
        lim = min(1, int(round(t / 3)))            # always 1 for t >= 3 (:395)
        reps = (target - 2 * lim) // (t - 2 * lim)
        mid = list(range(lim, t - lim)) * reps
        n_tail = target - len(mid) - lim
        return np.array(list(range(lim)) + mid + list(range(t - n_tail, t)), dtype=np.int32)
    return np.arange(target, dtype=np.int32)


def band_edges(n_rows, bands):
    """Row ranges of compress_bands(log=True) (util_audio.py:451-456)."""
    ind = np.geomspace(1, n_rows, bands + 1).astype(np.int64)
    ind[0] = 0
    for i in range(bands):
        sub = ind[i + 1] - ind[i]
        if sub < 1:
            ind[i + 1] += -sub + 1
    return ind.astype(np.int32)


# ======================================================================================
# Batched, device-resident windows
# ======================================================================================
_DEV_TABLES = {}


def _dev_table(key, build):
    """Small constant index tables live on the device once: a pageable host-to-device copy
    inside the loop would drain the stream on every call."""
    t = _DEV_TABLES.get(key)
    if t is None:
        t = _DEV_TABLES[key] = to_dev(build(), torch.int32)
    return t


class AudioBatch:
    """B equally long windows processed together.  State: wave [B, L] f32,
    mag [B, T, ldf] f32, ph [B, T, ldf, 2] f32 (unit complex), ref_max [B] f32."""

    def __init__(self, wave, n_fft, hop_length=None, center=True, sample_rate=44100):
        self.dev = require_gpu()
        self.N = int(n_fft)
        self.hl = int(hop_length) if hop_length is not None else int(math.floor(n_fft / 4))
        self.center = bool(center)
        self.sr = sample_rate
        self.F = self.N // 2 + 1
        self.ldf = ldf_of(self.N)
        self.plan = _plan(self.N, self.hl, self.center)
        self.lib = _lib.load()
        self.wave = None
        self.mag = self.ph = self.ref_max = None
        self._fmax = None            # (per-frame maxima [B, T], data_ptr of the mag they describe): compress_bands(fmax=True)
        self._max_scratch = None
        if wave is not None:
            w = to_dev(wave)
            if w.dim() == 1:
                w = w[None, :]
            self.wave = w.contiguous()
            self.B, self.L = self.wave.shape
            self.T = self.lib.amt_stft_frames(self.plan, int(self.L))
            if self.T < 0:
                _lib.check(self.T)

    # -- STFT / iSTFT --------------------------------------------------------------
    def stft(self, with_phase=True):
        """audio_complete.mag/.ph/.ref_mag for all windows (util_audio.py:139-174)."""
        B, T, ldf = self.B, self.T, self.ldf
        self._fmax = None
        self.mag = empty((B, T, ldf))
        self.ph = empty((B, T, ldf, 2)) if with_phase else None
        self.ref_max = empty((B,))
        _lib.check(self.lib.amt_stft_mag(
            self.plan, ptr(self.wave), B, int(self.L), _stride0(self.wave), ptr(self.mag),
            ptr(self.ph), ptr(self.ref_max), T, ldf, T * ldf, stream_ptr()))
        return self

    def istft(self):
        """audio_complete.wf from mag*ph (util_audio.py:94-97): [B, hop*(T-1)]."""
        B, T = self.mag.shape[0], self.mag.shape[1]
        lout = self.hl * (T - 1) if self.center else self.N + self.hl * (T - 1)
        out = empty((B, lout))
        _lib.check(self.lib.amt_istft(self.plan, ptr(self.mag), ptr(self.ph), B, T, self.ldf,
                                      T * self.ldf, ptr(out), lout, stream_ptr()))
        return out

    def window_max(self):
        out = empty((self.mag.shape[0],))
        T = self.mag.shape[1]
        _lib.check(self.lib.amt_window_max(ptr(self.mag), self.mag.shape[0], T, self.ldf,
                                           T * self.ldf, ptr(out), stream_ptr()))
        return out

    # -- subtraction -----------------------------------------------------------------
    def subtract(self, guess_mag, guess_max=None, guess_index=None, guess_frames=None,
                 offset_frames=None, normalize=True, relu=True, overkill_factor=1.0, span=False):
        """audio_complete.subtract for every window (util_audio.py:221-259).
        guess_mag [G, Tg, ldf] f32 frame-major; guess_index [B] int32 selects the
        guess per window; offset_frames [B] int32 is the already-clamped first frame
        max(_seconds_to_frames(offset) - attack_compensation, 0).
        span=True (the caller vouches that mag is non-negative and unchanged since the compress_bands(fmax=True) that
        left its per-frame maxima): amt_subtract_span touches only the frames the guess covers; same residual, same
        maxima, ~2.5 x fewer bytes.  Falls back to the whole-window kernel when the maxima are not there."""
        B, T = self.mag.shape[0], self.mag.shape[1]
        a = _lib.SubtractArgs()
        new_max = empty((B,))
        a.resid = self.mag.data_ptr()
        a.resid_max = self.ref_max.data_ptr() if normalize else None
        a.guess = guess_mag.data_ptr()
        a.guess_max = guess_max.data_ptr() if (normalize and guess_max is not None) else None
        a.guess_index = guess_index.data_ptr() if guess_index is not None else None
        if isinstance(guess_frames, torch.Tensor):
            a.guess_frames = guess_frames.data_ptr()
            a.guess_frames_all = 0
        else:
            a.guess_frames = None
            a.guess_frames_all = int(guess_mag.shape[1] if guess_frames is None else guess_frames)
        a.offset_frames = offset_frames.data_ptr() if offset_frames is not None else None
        a.new_max = new_max.data_ptr()
        a.resid_stride = T * self.ldf
        a.guess_stride = _stride0(guess_mag)
        a.B, a.T, a.ldf, a.F = B, T, self.ldf, self.F
        a.normalize = int(bool(normalize))
        a.relu = int(bool(relu))
        a.overkill_factor = float(overkill_factor)
        fm = self._fmax
        # the cached maxima describe `mag` only while it is the same tensor AND unedited since (the key holds its address
        # and torch's in-place version counter; the kernels behind this class write through raw pointers, which the counter
        # does not see -- those writers are this class's own and refresh or drop the cache themselves)
        if span and relu and fm is not None and fm[1] == (self.mag.data_ptr(), self.mag._version) and fm[0].shape == (B, T):
            _lib.check(self.lib.amt_subtract_span(C.byref(a), ptr(fm[0]), int(guess_mag.shape[1]), stream_ptr()))
        else:
            self._fmax = None
            _lib.check(self.lib.amt_subtract(C.byref(a), stream_ptr()))
        self.ref_max = new_max
        return self

    # -- features -------------------------------------------------------------------
    def compress_bands(self, bands, ref=None, target_frames=None, fmax=False):
        """_resize(compress_bands(mag, bands), target)/ref -> [B, bands, target]
        (training.py:333-336).  fmax=True (identity frame map only): the kernel also leaves every frame's maximum, for
        subtract(span=True)."""
        B, T = self.mag.shape[0], self.mag.shape[1]
        target = T if target_frames is None else int(target_frames)
        edges = _dev_table(('edges', self.F, bands), lambda: band_edges(self.F, bands))
        src = None if target == T else _dev_table(('resize', T, target),
                                                  lambda: resize_source_frames(T, target))
        out = empty((B, bands, target))
        if fmax and src is None:
            fm = empty((B, T))
            _lib.check(self.lib.amt_compress_bands_fmax(
                ptr(self.mag), B, T, self.F, self.ldf, T * self.ldf, ptr(edges), bands, ptr(ref),
                None, ptr(out), target, ptr(fm), stream_ptr()))
            self._fmax = (fm, (self.mag.data_ptr(), self.mag._version))
            return out
        _lib.check(self.lib.amt_compress_bands(
            ptr(self.mag), B, T, self.F, self.ldf, T * self.ldf, ptr(edges), bands, ptr(ref),
            ptr(src), ptr(out), target, stream_ptr()))
        return out

    def short_window(self, src_frame, band_min, bands, ref=None, mode=0):
        """resize(['mag','ph']) + section_power + scaling (training.py:337-363):
        [B, bands, frames]."""
        B, T = self.mag.shape[0], self.mag.shape[1]
        frames = src_frame.shape[1]
        out = empty((B, bands, frames))
        _lib.check(self.lib.amt_short_window(
            ptr(self.mag), ptr(self.ph), B, T, self.F, self.ldf, T * self.ldf, ptr(src_frame),
            frames, ptr(band_min), bands, ptr(ref), int(mode), ptr(out), stream_ptr()))
        return out


def gather_frames(src, index, F, elem=1, band_min=0, bands=None):
    """Frame / band selection on the device (amt_gather_frames).  src: device [T, ldf] (elem 1) or
    [T, ldf, 2] (elem 2), or batched with a leading B; index: int sequence or device int32 tensor of
    source frames (-1 = zero frame), one table for all windows.  Returns [n, ldf_out(, 2)] (or batched),
    ldf_out = `bands` rounded up to 4; bins outside [0, F) and frames outside [0, T) read as zero."""
    lib = _lib.load()
    batched = src.dim() == (3 if elem == 1 else 4)
    s = src if batched else src[None]
    B, T, ldf = s.shape[0], s.shape[1], s.shape[2]
    bands = F if bands is None else int(bands)
    ldo = (bands + 3) & ~3
    idx = index if isinstance(index, torch.Tensor) else to_dev(np.asarray(index, dtype=np.int32), torch.int32)
    n = int(idx.shape[-1])
    shape = (B, n, ldo) if elem == 1 else (B, n, ldo, 2)
    out = empty(shape)
    if n:
        if T == 0:
            out.zero_()
        else:
            _lib.check(lib.amt_gather_frames(ptr(s), B, T, int(F), ldf, T * ldf * elem, elem, ptr(idx), 0, n,
                                             int(band_min), bands, ptr(out), ldo, n * ldo * elem, stream_ptr()))
    return out if batched else out[0]


def amplitude_to_db(mag, ref, F, window_max=None, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db on the device for [B, T, ldf] magnitudes; ref / window_max: [B] tensors."""
    lib = _lib.load()
    B, T, ldf = mag.shape
    if window_max is None:
        window_max = empty((B,))
        _lib.check(lib.amt_window_max(ptr(mag), B, T, ldf, T * ldf, ptr(window_max), stream_ptr()))
    out = empty((B, T, ldf))
    _lib.check(lib.amt_amplitude_to_db(ptr(mag), B, T, int(F), ldf, T * ldf, ptr(ref), ptr(window_max),
                                       float(amin), -1.0 if top_db is None else float(top_db), ptr(out),
                                       stream_ptr()))
    return out


def db_to_amplitude(db, ref, F):
    lib = _lib.load()
    B, T, ldf = db.shape
    out = empty((B, T, ldf))
    _lib.check(lib.amt_db_to_amplitude(ptr(db), B, T, int(F), ldf, T * ldf, ptr(ref), ptr(out), stream_ptr()))
    return out


def cqt_slices(wave, src_frame, table, n_bins, hop, bin0=None, ref=None, complex_out=False):
    """slice_C for a batch (util_audio.py:411-434, build-defined CQT):
    wave [B, L] f32 device; src_frame [B, frames] int32; table = cqt_table(..., device) =
    (phase_inc, length, coef).  Returns [B, n_bins, frames]; with complex_out the pair (real, imaginary) of that
    shape, phase referred to each frame's centre (util_audio.py:428: magnitude_only=False keeps librosa's complex C)."""
    lib = _lib.load()
    B, L = wave.shape
    frames = src_frame.shape[1]
    out = empty((B, n_bins, frames))
    a = _lib.CqtArgs()
    a.wave = wave.data_ptr()
    a.src_frame = src_frame.data_ptr()
    a.bin0 = bin0.data_ptr() if bin0 is not None else None
    a.phase_inc = table[0].data_ptr()
    a.length = table[1].data_ptr()
    a.ref = ref.data_ptr() if ref is not None else None
    a.coef = table[2].data_ptr()
    a.out = out.data_ptr()
    a.wave_stride = _stride0(wave)
    a.B, a.L, a.hop, a.frames, a.n_bins, a.n_table = B, L, int(hop), frames, int(n_bins), \
        int(table[0].shape[0])
    if complex_out:
        out_im = empty((B, n_bins, frames))
        _lib.check(lib.amt_cqt_slices_complex(C.byref(a), ptr(out_im), stream_ptr()))
        return out, out_im
    _lib.check(lib.amt_cqt_slices(C.byref(a), stream_ptr()))
    return out


_MFMA_TABLES = {}          # (phase_inc ptr, length ptr, n_bins, hop) -> (device table, the tensors it was built from)


def cqt_window_max(wave, table, hop, form='auto'):
    """max over every bin of `table` and every STFT-grid frame 0 .. L // hop of the signals' CQT: the song-level
    normalisers np.max(slice_C(0, duration, n_frames, ...)) of training.py:271-282.  wave [B, L] f32 device --
    windows, or a whole song as one row; table from cqt_table(..., device).  Returns [B].
    form: 'auto' = the MFMA form (one GEMM per window against a phasor table, amt_cqt_window_max_mfma) where its
    geometry fits -- batches of windows of at most 544 hop-blocks -- else the O(L)-per-bin VALU form
    (amt_cqt_window_max: any length, a whole song included); 'valu' / 'mfma' force one."""
    lib = _lib.load()
    B, L = wave.shape
    n_bins = int(table[0].shape[0])
    out = empty((B,))
    if form not in ('auto', 'valu', 'mfma'):
        raise ValueError('Requested attribute does not exist')
    if form != 'valu':
        key = (table[0].data_ptr(), table[1].data_ptr(), n_bins, int(hop))
        ent = _MFMA_TABLES.get(key)
        nbytes = int(lib.amt_cqt_mfma_table_bytes(int(hop), n_bins))
        if ent is None and nbytes:
            tab = empty(((nbytes + 3) // 4,), torch.int32)
            _lib.check(lib.amt_cqt_mfma_table(ptr(table[0]), ptr(table[1]), n_bins, int(hop), ptr(tab), stream_ptr()))
            ent = _MFMA_TABLES[key] = (tab, table[0], table[1])       # keeps the source tensors (and their addresses) alive
        if ent is not None:
            scratch = empty((B,))
            st = lib.amt_cqt_window_max_mfma(ptr(wave), B, L, _stride0(wave), int(hop), ptr(table[0]), ptr(table[1]),
                                             ptr(ent[0]), n_bins, ptr(out), ptr(scratch), stream_ptr())
            if st == _lib.AMT_OK:
                return out
            if st != _lib.AMT_E_UNSUPPORTED or form == 'mfma':
                _lib.check(st)
        elif form == 'mfma':
            _lib.check(_lib.AMT_E_UNSUPPORTED)
    need = int(lib.amt_cqt_window_max_workspace(L, int(hop), n_bins, B))
    ws = empty(((need + 7) // 8,), torch.float64) if need else None
    _lib.check(lib.amt_cqt_window_max(ptr(wave), B, L, _stride0(wave), int(hop), ptr(table[0]), ptr(table[1]),
                                      ptr(table[2]), n_bins, ptr(out), ptr(ws) if need else None, need,
                                      stream_ptr()))
    return out


FILTER_SCALE = 2.0     # slice_C hard-codes filter_scale=2 (util_audio.py:426)


def cqt_table(sr, fmin_hz, n_bins, bins_per_octave, device=None):
    """Per-bin oscillator increment (uint32 cycles/sample * 2^32) and filter
    length N_k = ceil(Q sr / f_k), Q = 2/(2^(1/bpo)-1)."""
    k = np.arange(n_bins, dtype=np.float64)
    freq = float(fmin_hz) * 2.0 ** (k / bins_per_octave)
    q = FILTER_SCALE / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    length = np.ceil(q * sr / freq).astype(np.int32)
    inc = (np.rint(freq / sr * 2.0 ** 32).astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    if device is None:
        return inc, length
    # device form: (phase_inc, length, coef) -- coef = the per-bin phasor table the CQT kernels read through the
    # scalar cache (amt_cqt_coef), built once here
    inc_d, len_d = torch.from_numpy(inc.view(np.int32)).to(device), torch.from_numpy(length).to(device)
    coef = empty((int(n_bins), 192))
    _lib.check(_lib.load().amt_cqt_coef(ptr(inc_d), ptr(len_d), int(n_bins), ptr(coef), stream_ptr()))
    return (inc_d, len_d, coef)


_NOTE = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}


def note_to_midi(note):
    if isinstance(note, (int, np.integer)):
        return int(note)
    name, rest, acc = note[0].upper(), note[1:], 0
    while rest and rest[0] in '#b':
        acc += 1 if rest[0] == '#' else -1
        rest = rest[1:]
    return 12 * ((int(rest) if rest else 0) + 1) + _NOTE[name] + acc


# ======================================================================================
# Reference call surface: one window
# ======================================================================================
def _to_tm(a, ldf, complex_=False):
    """numpy [F, T] -> device frame-major [T, ldf] (float) or [T, ldf, 2]."""
    a = np.asarray(a)
    Fb, T = a.shape
    if complex_:
        a = a.astype(np.complex64)
        h = np.zeros((T, ldf, 2), dtype=np.float32)
        h[:, :Fb, 0] = a.real.T
        h[:, :Fb, 1] = a.imag.T
    else:
        h = np.zeros((T, ldf), dtype=np.float32)
        h[:, :Fb] = a.T
    return to_dev(h)


def _from_tm(t, Fb, complex_=False):
    h = t.detach().cpu().numpy()
    if complex_:
        return np.ascontiguousarray((h[:, :Fb, 0] + 1j * h[:, :Fb, 1]).astype(np.complex64).T)
    return np.ascontiguousarray(h[:, :Fb].T)


SPECTRAL = ('mag', 'ph', 'D')          # attributes with a device copy; elem floats per bin: 1, 2, 1
_ELEM = {'mag': 1, 'ph': 2, 'D': 1}


class audio_complete:
    """Same constructor, properties and methods as util_audio.audio_complete (util_audio.py:32-527).

    Every spectral attribute (mag, ph, D) has two faces: a device tensor in the frame-major layout
    (``_d[name]``: [T, ldf] or [T, ldf, 2]), which all numeric work reads and writes, and a host
    array in the reference's [F, T] layout (``_h[name]``), materialised only when the caller looks
    at it or hands one in.  Window management (section / slice / concat / resize / section_power) is
    expressed as frame index maps executed by amt_gather_frames -- a column range of the
    reference's arrays is a row range here, zero padding is the index -1 -- so a window never
    travels to the host to be cut."""

    def __init__(self, waveform, n_fft, hop_length=None, center=True, sample_rate=44100):
        self._wf = waveform
        self._F = None
        self._h = {k: None for k in SPECTRAL}       # host views, [F, T]
        self._d = {k: None for k in SPECTRAL}       # device tensors, frame-major
        self._ref_mag = None
        self.sr = sample_rate
        self.N = n_fft
        self.center = center
        self.hl = hop_length if hop_length is not None else int(np.floor(n_fft / 4))
        self._fft_freq = fft_frequencies(sample_rate, n_fft)

    # ---- the two faces of a spectral attribute --------------------------------------
    @property
    def _Fb(self):
        return self.N // 2 + 1

    @property
    def _ldf(self):
        return ldf_of(self.N)

    def _has(self, name):
        return self._h[name] is not None or self._d[name] is not None

    def _dev(self, name):
        """Device tensor of a present attribute (uploads a host-set array once)."""
        if self._d[name] is None and self._h[name] is not None:
            self._d[name] = _to_tm(self._h[name], self._ldf, complex_=(name == 'ph'))
        return self._d[name]

    def _host(self, name):
        if self._h[name] is None and self._d[name] is not None:
            self._h[name] = _from_tm(self._d[name], self._Fb, complex_=(name == 'ph'))
        return self._h[name]

    def _put(self, name, host=None, dev=None):
        self._h[name], self._d[name] = host, dev

    def _frames_of(self, name):
        if self._d[name] is not None:
            return int(self._d[name].shape[0])
        return int(self._h[name].shape[1])

    # the reference's private fields, for callers (and tests) that poke them directly
    _mag = property(lambda self: self._host('mag'), lambda self, v: self._put('mag', host=v))
    _ph = property(lambda self: self._host('ph'), lambda self, v: self._put('ph', host=v))
    _D = property(lambda self: self._host('D'), lambda self, v: self._put('D', host=v))

    def _batch(self):
        return AudioBatch(None, self.N, self.hl, self.center, self.sr)

    def _scalar(self, v):
        return to_dev(np.array([v], dtype=np.float32))

    def _run_stft(self):
        wf = np.asarray(self.wf, dtype=np.float32)
        b = AudioBatch(wf[None, :], self.N, self.hl, self.center, self.sr).stft(True)
        self._put('mag', dev=b.mag[0])
        self._put('ph', dev=b.ph[0])

    def _mag_from_db(self):
        """mag = db_to_amplitude(D, ref_mag or 1) (util_audio.py:99-101,122-124,143-145), on the device."""
        if self._ref_mag is None:
            self._ref_mag = 1.0
        m = db_to_amplitude(self._dev('D')[None], self._scalar(self._ref_mag), self._Fb)[0]
        self._put('mag', dev=m)

    def clone(self):                                        # util_audio.py:69-87
        ac = audio_complete(None if self._wf is None else np.array(self._wf, copy=True),
                            sample_rate=self.sr, n_fft=self.N, center=self.center,
                            hop_length=self.hl)
        ac._F = None if self._F is None else np.array(self._F, copy=True)
        for k in SPECTRAL:
            ac._put(k, None if self._h[k] is None else np.array(self._h[k], copy=True),
                    None if self._d[k] is None else self._d[k].clone())
        ac._ref_mag = self._ref_mag
        return ac

    # ---- properties ---------------------------------------------------------------
    @property
    def wf(self):                                           # util_audio.py:88-106
        if self._wf is None:
            if self._F is not None:
                self._wf = self._istft_complex(self._F)
            elif self._has('mag') and self._has('ph'):
                self._wf = self._istft_magph()
            elif self._has('D') and self._has('ph'):
                self._mag_from_db()
                self._wf = self._istft_magph()
        return self._wf

    @wf.setter
    def wf(self, value):                                    # util_audio.py:107-114
        for k in SPECTRAL:
            self._put(k)
        self._ref_mag = None
        self._F = None
        self._wf = value

    def _istft_magph(self):
        b = self._batch()
        b.mag = self._dev('mag')[None]
        b.ph = self._dev('ph')[None]
        return b.istft()[0].cpu().numpy()

    def _istft_complex(self, Fc):
        b = self._batch()
        b.mag = _to_tm(Fc, self._ldf, True)[None]     # interleaved complex, pitch 2*ldf
        b.ph = None
        return b.istft()[0].cpu().numpy()

    @property
    def F(self):                                            # util_audio.py:116-129
        if self._F is None:
            if not (self._has('mag') and self._has('ph')):
                if self._has('D') and self._has('ph'):
                    self._mag_from_db()
                elif self.wf is not None:
                    self._run_stft()
            if self._has('mag') and self._has('ph'):
                self._F = self.mag * self.ph
        return self._F

    @F.setter
    def F(self, value):                                     # util_audio.py:130-137
        for k in SPECTRAL:
            self._put(k)
        self._ref_mag = None
        self._F = value
        self._wf = None

    def _magphase_of_F(self):
        # librosa.core.magphase of a caller-supplied F (util_audio.py:147): host numpy, off the hot path
        # (the hot path gets mag and ph straight from the STFT kernel)
        Fc = np.asarray(self._F)
        self._put('mag', host=np.abs(Fc))
        self._put('ph', host=np.exp(1.j * np.angle(Fc)).astype(Fc.dtype if np.iscomplexobj(Fc) else np.complex64))

    @property
    def mag(self):                                          # util_audio.py:139-148
        if not self._has('mag'):
            if self._has('D') and self._has('ph'):
                self._mag_from_db()
            elif self._F is not None:
                self._magphase_of_F()
            else:
                self._run_stft()
        return self._host('mag')

    @mag.setter
    def mag(self, val):                                     # util_audio.py:149-157
        self._put('D')
        self._ref_mag = None
        self._put('mag', host=val)
        if self._has('ph') and self.ph.shape != val.shape:
            self._put('ph')
        self._F = None
        self._wf = None

    @property
    def ph(self):                                           # util_audio.py:159-163
        if not self._has('ph'):
            if self._F is not None:
                self._magphase_of_F()
            else:
                self._run_stft()
        return self._host('ph')

    @ph.setter
    def ph(self, val):                                      # util_audio.py:164-168
        self._put('ph', host=val)
        self._F = None
        self._wf = None

    @property
    def ref_mag(self):                                      # util_audio.py:170-174
        if self._ref_mag is None:
            self.mag
            b = self._batch()
            b.mag = self._dev('mag')[None]
            self._ref_mag = np.float32(b.window_max()[0].item())
        return self._ref_mag

    @property
    def D(self):                                            # util_audio.py:176-180
        if not self._has('D'):
            self.mag
            m = self._dev('mag')[None]
            d = amplitude_to_db(m, self._scalar(self.ref_mag), self._Fb)      # max(D) from the window's own maximum
            self._put('D', dev=d[0])
        return self._host('D')

    @D.setter
    def D(self, val):                                       # util_audio.py:181-190
        self._put('D', host=val)
        if self._has('ph') and self.ph.shape != np.shape(val):
            self._put('ph')
        self._put('mag')
        self._F = None
        self._wf = None

    def _P(self, name):                                     # util_audio.py:192-207
        if name == 'wf':
            return self._wf
        if name == 'F':
            return self._F
        if name in SPECTRAL:
            return self._host(name)
        raise ValueError('Requested attribute does not exist')

    @property
    def shape(self):                                        # util_audio.py:209-218
        for k in SPECTRAL:
            if self._h[k] is not None:
                return tuple(self._h[k].shape)
            if self._d[k] is not None:
                return (self._Fb, int(self._d[k].shape[0]))
        return self.F.shape

    def _wf_len(self):
        """len(self.wf) without forcing the iSTFT the reference triggers here
        (util_audio.py:264): librosa.istft returns hop*(T-1) samples."""
        if self._wf is not None:
            return len(self._wf)
        T = self.shape[1]
        return self.hl * (T - 1) if self.center else self.N + self.hl * (T - 1)

    # ---- the subtraction step --------------------------------------------------
    def subtract(self, subtrahend, offset=0, attack_compensation=0,
                 normalize=True, relu=True, overkill_factor=1):
        """util_audio.py:221-259, executed by amt_subtract."""
        if isinstance(subtrahend, audio_complete):
            subtrahend.mag
            gmag = subtrahend._dev('mag')
            gmax = subtrahend.ref_mag if normalize else None
        else:
            g = np.asarray(subtrahend)
            gmag = _to_tm(g, self._ldf)
            gmax = np.float32(np.max(g)) if normalize else None
        self.mag
        T = self.shape[1]
        off = max(self._seconds_to_frames(offset) - attack_compensation, 0)
        if off > T:
            # the reference fails here with a negative np.zeros dimension (:250-257)
            raise ValueError('negative dimensions are not allowed')
        b = self._batch()
        b.mag = self._dev('mag')[None]                   # in place, as the reference's `self.mag -= ...`
        b.ref_max = self._scalar(self.ref_mag if normalize else 0.0)
        gm = self._scalar(gmax if normalize else 1.0)
        offs = to_dev(np.array([off], dtype=np.int32), torch.int32)
        b.subtract(gmag[None], gm, None, None, offs, normalize, relu, overkill_factor)
        # mag setter semantics (:149-157): dependants cleared, phase kept (same shape)
        self._put('D')
        self._put('mag', dev=b.mag[0])
        self._F = None
        self._wf = None
        self._ref_mag = None             # re-evaluated lazily, as np.max(self.mag) is in the reference

    def _seconds_to_frames(self, time):                     # util_audio.py:261-264
        return int(np.floor(time * self.shape[1] * self.sr / self._wf_len()))

    def _frames_to_seconds(self, frames):                   # util_audio.py:269-272
        return frames / self.shape[1] / self.sr * self._wf_len()

    def midi_tone_to_FFT(self, tone):                       # util_audio.py:278-284
        """Two bins below the first FFT bin above the tone, not below bin 0."""
        return max(bisect.bisect_right(self._fft_freq, midi_to_hz(tone)) - 2, 0)

    # ---- window management: frame index maps on the device -------------------------------
    def _take(self, index, names=SPECTRAL, with_F=True):
        """{attribute: selected frames} for every present attribute.  index[j] = source frame of
        output frame j, -1 = a frame of zeros.  Device attributes go through amt_gather_frames."""
        index = np.asarray(index, dtype=np.int32)
        out = {}
        for k in names:
            if self._has(k):
                out[k] = gather_frames(self._dev(k), index, self._Fb, _ELEM[k])
        if with_F and self._F is not None:
            Fc = np.asarray(self._F)
            sel = np.where(index[None, :] >= 0, Fc[:, np.clip(index, 0, max(Fc.shape[1] - 1, 0))], 0) \
                if Fc.shape[1] else np.zeros((Fc.shape[0], len(index)), Fc.dtype)
            out['F'] = sel
        return out

    def _wave_span(self, first, last):
        return (int(np.floor(self._frames_to_seconds(first) * self.sr)),
                int(np.floor(self._frames_to_seconds(last) * self.sr)))

    def section(self, start, end, duration_in_frames=None):  # util_audio.py:286-328
        """A copy of [start, end) seconds (or `duration_in_frames` frames from start); frames past the
        end of the spectra are zeros."""
        first = self._seconds_to_frames(start)
        last = self._seconds_to_frames(end) if duration_in_frames is None else first + duration_in_frames
        wave = None
        if self._wf is not None:
            a, b = self._wave_span(first, last)
            wave = np.array(self._wf[a:b], copy=True)
            if len(wave) < b - a:                           # the reference pads by (b - len), not (b - a - len)
                wave = np.concatenate((wave, np.zeros(b - len(wave))))
        cut = audio_complete(wave, self.N, hop_length=self.hl, center=self.center, sample_rate=self.sr)
        cut._ref_mag = self._ref_mag
        if any(self._has(k) for k in SPECTRAL) or self._F is not None:
            T = self.shape[1]
            # columns first:last that exist, then one zero frame for every frame `last` lies past the end
            index = list(range(T))[first:last] + [-1] * max(last - T, 0)
            for k, v in self._take(index).items():
                if k == 'F':
                    cut._F = v
                else:
                    cut._put(k, dev=v)
        return cut

    def section_power(self, name, band_min, band_max):      # util_audio.py:334-349
        """Rows [band_min, band_max) of an attribute, zero rows past the last bin: [bands, T]."""
        if name in SPECTRAL and self._has(name):
            T = self._frames_of(name)
            g = gather_frames(self._dev(name), np.arange(T), self._Fb, _ELEM[name], band_min, band_max - band_min)
            return _from_tm(g, band_max - band_min, complex_=(name == 'ph'))
        P = self._P(name)                                   # 'F' / 'wf' (host arrays), or an absent attribute
        rows = np.zeros((band_max - band_min,) + P.shape[1:], dtype=P.dtype)
        got = P[band_min:band_max]
        rows[:got.shape[0]] = got
        return rows

    def slice(self, start_in_frames, end_in_frames):        # util_audio.py:351-365
        """Keep frames [start, end) in place.  A frame range of the frame-major device tensors is a
        contiguous row range: a view, nothing is copied."""
        keep = slice(start_in_frames, end_in_frames)
        if self._wf is not None:
            self._wf = self._wf[int(self._frames_to_seconds(start_in_frames) * self.sr):
                                int(self._frames_to_seconds(end_in_frames) * self.sr)]
        if self._F is not None:
            self._F = self._F[:, keep]
        for k in SPECTRAL:
            if self._d[k] is not None:
                self._put(k, None if self._h[k] is None else self._h[k][:, keep], self._d[k][keep])
            elif self._h[k] is not None:
                self._put(k, host=self._h[k][:, keep])

    def concat(self, ac):                                   # util_audio.py:374-382
        """Append another window; an attribute only one side has is dropped (None)."""
        both = lambda a, b: a is not None and b is not None
        self._wf = np.concatenate((self._wf, ac._wf), axis=0) if both(self._wf, ac._wf) else None
        self._F = np.concatenate((self._F, ac._F), axis=1) if both(self._F, ac._F) else None
        for k in SPECTRAL:
            if self._has(k) and ac._has(k):
                a, b = self._dev(k), ac._dev(k)
                joined = empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]))
                for part, at in ((a, 0), (b, a.shape[0])):
                    if part.shape[0]:
                        joined[at:at + part.shape[0]] = gather_frames(part, np.arange(part.shape[0]),
                                                                      self._Fb, _ELEM[k])
                self._put(k, dev=joined)
            else:
                self._put(k)

    @staticmethod
    def _resize(P, target_frame_count):                     # util_audio.py:384-409
        P = np.asarray(P)
        t = P.shape[1]
        if t == 0:
            return np.zeros((P.shape[0], target_frame_count))
        if t == target_frame_count:
            return P
        return P[:, resize_source_frames(t, target_frame_count)]

    def slice_C(self, start, duration, target_frame_count, magnitude_only=True,
                bins_per_tone=1, filter_scale=2, highest_note='C8', lowest_note='A0',
                nbins=None):
        """util_audio.py:411-434 with the build-defined CQT (oracle/cqt.py); like the reference, `filter_scale` is
        ignored (:426).  magnitude_only=False returns the complex CQT (:428 skips the np.abs), complex64, its phase
        referred to each frame's centre."""
        if nbins is None:
            nbins = int((note_to_midi(highest_note) - note_to_midi(lowest_note)) * bins_per_tone)
        fmin = float(midi_to_hz(note_to_midi(lowest_note)))
        dev = require_gpu()
        wf = to_dev(np.asarray(self.wf, dtype=np.float32))[None]
        src = self._resize_index(start, duration, target_frame_count)
        table = cqt_table(self.sr, fmin, nbins, int(12 * bins_per_tone), dev)
        out = np.zeros((nbins, target_frame_count), dtype=np.float32 if magnitude_only else np.complex64)
        for c0 in range(0, target_frame_count, 8):
            cols = src[c0:c0 + 8]
            o = cqt_slices(wf, to_dev(cols[None], torch.int32), table, nbins, self.hl, complex_out=not magnitude_only)
            if magnitude_only:
                out[:, c0:c0 + len(cols)] = o[0].cpu().numpy()
            else:
                out[:, c0:c0 + len(cols)] = o[0][0].cpu().numpy() + 1j * o[1][0].cpu().numpy()
        return out

    @staticmethod
    def compress_bands(spectrum, bands=80, log=True):       # util_audio.py:436-466
        spectrum = np.asarray(spectrum)
        Fb, T = spectrum.shape
        lib = _lib.load()
        if log:
            edges = band_edges(Fb, bands)
        else:
            r = Fb // bands
            edges = (np.arange(bands + 1) * r).astype(np.int32)
        ldf = (Fb + 3) & ~3
        d = _to_tm(spectrum, ldf)
        out = empty((1, bands, T))
        _lib.check(lib.amt_compress_bands(ptr(d), 1, T, Fb, ldf, T * ldf,
                                          ptr(to_dev(edges, torch.int32)), bands, None, None,
                                          ptr(out), T, stream_ptr()))
        return out[0].cpu().numpy().astype(np.float64)

    def _resize_index(self, start, duration, target):
        """Source frame of every column of _resize(X[:, s:t], target) (util_audio.py:384-409), -1 = zeros."""
        T = self.shape[1]
        cols = np.arange(T)[self._seconds_to_frames(start):self._seconds_to_frames(start + duration)]
        rel = resize_source_frames(len(cols), target)
        return np.where(rel < 0, -1, cols[np.clip(rel, 0, max(len(cols) - 1, 0))] if len(cols) else -1).astype(np.int32)

    def resize(self, start, duration, target_frame_count, attribs=['F']):
        """util_audio.py:469-507: a new object holding `target_frame_count` frames of the requested
        attributes, cut from [start, start + duration) by the tile / crop rule of _resize."""
        short = audio_complete(None, self.N, hop_length=self.hl, center=self.center, sample_rate=self.sr)
        if self._ref_mag is not None:
            short._ref_mag = self._ref_mag
        index = None
        for attrib in attribs:
            if attrib not in ('F', 'mag', 'ph', 'D'):
                raise ValueError('Invalid attribute requested')
            getattr(self, attrib)                           # materialise it (STFT on first use)
            if index is None:
                index = self._resize_index(start, duration, target_frame_count)
            if attrib == 'F':
                short.F = self._take(index, names=())['F']
            else:
                got = self._take(index, names=(attrib,), with_F=False)[attrib]
                # what the reference's setter does on assignment (:149-157, :164-168, :181-190)
                if attrib == 'mag':
                    short._put('D'); short._ref_mag = None
                    if short._has('ph') and short._frames_of('ph') != target_frame_count:
                        short._put('ph')
                elif attrib == 'D':
                    if short._has('ph') and short._frames_of('ph') != target_frame_count:
                        short._put('ph')
                    short._put('mag')
                short._put(attrib, dev=got)
                short._F = None
                short._wf = None
        return short

    def spectral_flatness(self):                            # util_audio.py:330-332
        """np.mean(librosa.feature.spectral_flatness(y=wf, n_fft, hop)) (power = 2, amin = 1e-10): per frame the
        geometric over the arithmetic mean of max(amin, |STFT|^2).  training.py:266 skips a song whose render
        is white noise (> 0.3)."""
        lib = _lib.load()
        self.mag
        m = self._dev('mag')[None]
        out = empty((1, m.shape[1]))
        _lib.check(lib.amt_spectral_flatness(ptr(m), 1, m.shape[1], self._Fb, self._ldf, m.shape[1] * self._ldf,
                                             1e-10, ptr(out), stream_ptr()))
        return float(np.mean(out.cpu().numpy().astype(np.float64)))

    def save(self, filename, flac=True):
        """util_audio.py:520-527: the waveform as PCM-24 FLAC (the WAV branch is not provided)."""
        if not flac:
            raise NotImplementedError('only FLAC output is provided')
        from . import flac as _flac
        _flac.save_float(np.asarray(self.wf), filename, sr=self.sr, bps=24)
