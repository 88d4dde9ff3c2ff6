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


def ldf_of(n_fft):
    return ((n_fft // 2 + 1) + 3) & ~3


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
                 offset_frames=None, normalize=True, relu=True, overkill_factor=1.0):
        """audio_complete.subtract for every window (util_audio.py:221-259).
        guess_mag [G, Tg, ldf] f32 frame-major; guess_index [B] int32 selects the
        guess per window; offset_frames [B] int32 is the already-clamped first frame
        max(_seconds_to_frames(offset) - attack_compensation, 0)."""
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
        _lib.check(self.lib.amt_subtract(C.byref(a), stream_ptr()))
        self.ref_max = new_max
        return self

    # -- features -------------------------------------------------------------------
    def compress_bands(self, bands, ref=None, target_frames=None):
        """_resize(compress_bands(mag, bands), target)/ref -> [B, bands, target]
        (training.py:333-336)."""
        B, T = self.mag.shape[0], self.mag.shape[1]
        target = T if target_frames is None else int(target_frames)
        edges = _dev_table(('edges', self.F, bands), lambda: band_edges(self.F, bands))
        src = None if target == T else _dev_table(('resize', T, target),
                                                  lambda: resize_source_frames(T, target))
        out = empty((B, bands, target))
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


def cqt_slices(wave, src_frame, table, n_bins, hop, bin0=None, ref=None):
    """slice_C for a batch (util_audio.py:411-434, build-defined CQT):
    wave [B, L] f32 device; src_frame [B, frames] int32; table = (phase_inc uint32
    tensor, length int32 tensor).  Returns [B, n_bins, frames]."""
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
    a.out = out.data_ptr()
    a.wave_stride = _stride0(wave)
    a.B, a.L, a.hop, a.frames, a.n_bins, a.n_table = B, L, int(hop), frames, int(n_bins), \
        int(table[0].shape[0])
    _lib.check(lib.amt_cqt_slices(C.byref(a), stream_ptr()))
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
    return (torch.from_numpy(inc.view(np.int32)).to(device), torch.from_numpy(length).to(device))


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


class audio_complete:
    """Same constructor, properties and methods as util_audio.audio_complete."""

    def __init__(self, waveform, n_fft, hop_length=None, center=True, sample_rate=44100):
        self._wf = waveform
        self._F = None
        self._mag = None        # host views are materialised lazily from the device tensors
        self._ref_mag = None
        self._ph = None
        self._D = None
        self._dmag = None       # device [T, ldf]
        self._dph = None        # device [T, ldf, 2]
        self.sr = sample_rate
        self.N = n_fft
        self.center = center
        self.hl = hop_length if hop_length is not None else int(np.floor(n_fft / 4))
        self._fft_freq = fft_frequencies(sample_rate, n_fft)

    # ---- helpers ----------------------------------------------------------------
    @property
    def _Fb(self):
        return self.N // 2 + 1

    @property
    def _ldf(self):
        return ldf_of(self.N)

    def _batch(self):
        b = AudioBatch(None, self.N, self.hl, self.center, self.sr)
        return b

    def _run_stft(self):
        wf = np.asarray(self.wf, dtype=np.float32)
        b = AudioBatch(wf[None, :], self.N, self.hl, self.center, self.sr).stft(True)
        self._dmag, self._dph = b.mag[0], b.ph[0]
        self._mag = None
        self._ph = None

    def _have_mag(self):
        return self._mag is not None or self._dmag is not None

    def _have_ph(self):
        return self._ph is not None or self._dph is not None

    def _dev_mag(self):
        if self._dmag is None:
            self._dmag = _to_tm(self._mag, self._ldf)
        return self._dmag

    def _dev_ph(self):
        if self._dph is None:
            self._dph = _to_tm(self._ph, self._ldf, True)
        return self._dph

    def clone(self):                                        # util_audio.py:69-87
        ac = audio_complete(None if self._wf is None else np.array(self._wf, copy=True),
                            sample_rate=self.sr, n_fft=self.N, center=self.center,
                            hop_length=self.hl)
        ac._F = None if self._F is None else self._F.copy()
        ac._mag = None if self._mag is None else self._mag.copy()
        ac._ph = None if self._ph is None else self._ph.copy()
        ac._D = None if self._D is None else np.array(self._D, copy=True)
        ac._dmag = None if self._dmag is None else self._dmag.clone()
        ac._dph = None if self._dph is None else self._dph.clone()
        ac._ref_mag = self._ref_mag
        return ac

    # ---- properties ---------------------------------------------------------------
    @property
    def wf(self):                                           # util_audio.py:88-106
        if self._wf is None:
            if self._F is not None:
                self._wf = self._istft_complex(self._F)
            elif self._have_mag() and self._have_ph():
                self._wf = self._istft_magph()
            elif self._D is not None and self._have_ph():
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self.__set_mag_host(_db_to_amplitude(self._D, self._ref_mag))
                self._wf = self._istft_magph()
        return self._wf

    @wf.setter
    def wf(self, value):                                    # util_audio.py:107-114
        self._D = None
        self._ref_mag = None
        self._mag = self._dmag = None
        self._ph = self._dph = None
        self._F = None
        self._wf = value

    def __set_mag_host(self, m):
        self._mag = np.asarray(m)
        self._dmag = None

    def _istft_magph(self):
        b = self._batch()
        b.mag = self._dev_mag()[None]
        b.ph = self._dev_ph()[None]
        return b.istft()[0].cpu().numpy()

    def _istft_complex(self, Fc):
        b = self._batch()
        b.mag = _to_tm(Fc, self._ldf, True)[None]     # interleaved complex, pitch 2*ldf
        b.ph = None
        return b.istft()[0].cpu().numpy()

    @property
    def F(self):                                            # util_audio.py:116-129
        if self._F is None:
            if self._have_mag() and self._have_ph():
                self._F = self.mag * self.ph
            elif self._D is not None and self._have_ph():
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self.__set_mag_host(_db_to_amplitude(self._D, self._ref_mag))
                self._F = self.mag * self.ph
            elif self.wf is not None:
                self._run_stft()
                self._F = self.mag * self.ph
        return self._F

    @F.setter
    def F(self, value):                                     # util_audio.py:130-137
        self._D = None
        self._ref_mag = None
        self._mag = self._dmag = None
        self._ph = self._dph = None
        self._F = value
        self._wf = None

    def _magphase_of_F(self):
        # librosa.core.magphase on a user-supplied F (util_audio.py:147): host numpy,
        # not on the hot path (the hot path gets mag/ph straight from the STFT kernel)
        Fc = np.asarray(self._F)
        self._mag = np.abs(Fc)
        self._ph = np.exp(1.j * np.angle(Fc)).astype(Fc.dtype if np.iscomplexobj(Fc) else np.complex64)
        self._dmag = self._dph = None

    @property
    def mag(self):                                          # util_audio.py:139-148
        if not self._have_mag():
            if self._D is not None and self._have_ph():
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self.__set_mag_host(_db_to_amplitude(self._D, self._ref_mag))
            elif self._F is not None:
                self._magphase_of_F()
            else:
                self._run_stft()
        if self._mag is None:
            self._mag = _from_tm(self._dmag, self._Fb)
        return self._mag

    @mag.setter
    def mag(self, val):                                     # util_audio.py:149-157
        self._D = None
        self._ref_mag = None
        self._mag = val
        self._dmag = None
        if self._have_ph() and self.ph.shape != val.shape:
            self._ph = self._dph = None
        self._F = None
        self._wf = None

    @property
    def ph(self):                                           # util_audio.py:159-163
        if not self._have_ph():
            if self._F is not None:
                self._magphase_of_F()
            else:
                self._run_stft()
        if self._ph is None:
            self._ph = _from_tm(self._dph, self._Fb, True)
        return self._ph

    @ph.setter
    def ph(self, val):                                      # util_audio.py:164-168
        self._ph = val
        self._dph = None
        self._F = None
        self._wf = None

    @property
    def ref_mag(self):                                      # util_audio.py:170-174
        if self._ref_mag is None:
            if not self._have_mag():
                self.mag
            b = self._batch()
            b.mag = self._dev_mag()[None]
            self._ref_mag = np.float32(b.window_max()[0].item())
        return self._ref_mag

    @property
    def D(self):                                            # util_audio.py:176-180
        if self._D is None:
            self._D = _amplitude_to_db(self.mag, self.ref_mag)
        return self._D

    @D.setter
    def D(self, val):                                       # util_audio.py:181-190
        self._D = val
        if self._have_ph() and self.ph.shape != val.shape:
            self._ph = self._dph = None
        self._mag = self._dmag = None
        self._F = None
        self._wf = None

    def _P(self, name):                                     # util_audio.py:192-207
        if name == 'wf':
            return self._wf
        elif name == 'F':
            return self._F
        elif name == 'mag':
            return self.mag if self._have_mag() else None
        elif name == 'ph':
            return self.ph if self._have_ph() else None
        elif name == 'D':
            return self._D
        raise ValueError('Requested attribute does not exist')

    @property
    def shape(self):                                        # util_audio.py:209-218
        if self._mag is not None:
            return self._mag.shape
        if self._dmag is not None:
            return (self._Fb, self._dmag.shape[0])
        if self._ph is not None:
            return self._ph.shape
        if self._dph is not None:
            return (self._Fb, self._dph.shape[0])
        if self._D is not None:
            return self._D.shape
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
            gmag = subtrahend._dev_mag()
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
        b.mag = self._dev_mag()[None]
        b.ref_max = to_dev(np.array([self.ref_mag if normalize else 0.0], dtype=np.float32))
        gm = to_dev(np.array([gmax if normalize else 1.0], dtype=np.float32))
        offs = to_dev(np.array([off], dtype=np.int32), torch.int32)
        b.subtract(gmag[None], gm, None, None, offs, normalize, relu, overkill_factor)
        # mag setter semantics (:149-157): dependants cleared, phase kept (same shape)
        self._D = None
        self._mag = None
        self._dmag = b.mag[0]
        self._F = None
        self._wf = None
        self._ref_mag = None             # re-evaluated lazily, as np.max(self.mag) is in the reference

    def _seconds_to_frames(self, time):                     # util_audio.py:261-264
        return int(np.floor(time * self.shape[1] * self.sr / self._wf_len()))

    def _frames_to_seconds(self, frames):                   # util_audio.py:269-272
        return frames / self.shape[1] / self.sr * self._wf_len()

    def midi_tone_to_FFT(self, tone):                       # util_audio.py:278-284
        f = midi_to_hz(tone)
        ind = bisect.bisect_right(self._fft_freq, f) - 1
        ind = 0 if ind == 0 else ind - 1
        return ind

    # ---- window management -----------------------------------------------------
    def section(self, start, end, duration_in_frames=None):  # util_audio.py:286-328
        tfs = self._seconds_to_frames(start)
        if duration_in_frames is None:
            tfe = self._seconds_to_frames(end)
        else:
            tfe = tfs + duration_in_frames
        if self._wf is not None:
            wav_start = int(np.floor(self._frames_to_seconds(tfs) * self.sr))
            wav_end = int(np.floor(self._frames_to_seconds(tfe) * self.sr))
            wav_cp = np.array(self._wf[wav_start:wav_end], copy=True)
            if wav_cp.shape[0] < wav_end - wav_start:
                wav_cp = np.concatenate((wav_cp, np.zeros(wav_end - wav_cp.shape[0])))
        else:
            wav_cp = None
        nac = audio_complete(wav_cp, self.N, hop_length=self.hl, center=self.center,
                             sample_rate=self.sr)

        def cc(f):
            if f is not None:
                cpd = np.array(f[:, tfs:tfe], copy=True)
                if f.shape[1] >= tfe:
                    return cpd
                return np.concatenate((cpd, np.zeros((f.shape[0], tfe - f.shape[1]))), axis=1)

        nac._F = cc(self._F)
        nac._ref_mag = self._ref_mag
        nac._mag = cc(self.mag if self._have_mag() else None)
        nac._ph = cc(self.ph if self._have_ph() else None)
        nac._D = cc(self._D)
        return nac

    def section_power(self, name, band_min, band_max):      # util_audio.py:334-349
        P = self._P(name)
        h = P.shape[0]
        cpd = np.array(P[band_min:band_max, :], copy=True)
        if band_max > h:
            cpd = np.concatenate((cpd, np.zeros((band_max - h, P.shape[1]))), axis=0)
        return cpd

    def slice(self, start_in_frames, end_in_frames):        # util_audio.py:351-365
        if self._wf is not None:
            self._wf = self._wf[int(self._frames_to_seconds(start_in_frames) * self.sr):
                                int(self._frames_to_seconds(end_in_frames) * self.sr)]
        if self._F is not None:
            self._F = self._F[:, start_in_frames:end_in_frames]
        if self._have_mag():
            self._mag = self.mag[:, start_in_frames:end_in_frames]
            self._dmag = None
        if self._have_ph():
            self._ph = self.ph[:, start_in_frames:end_in_frames]
            self._dph = None
        if self._D is not None:
            self._D = self._D[:, start_in_frames:end_in_frames]

    @staticmethod
    def _concus(dest, src, axis=1):                         # util_audio.py:368-372
        if src is None or dest is None:
            return None
        return np.concatenate((dest, src), axis=axis)

    def concat(self, ac):                                   # util_audio.py:374-382
        self._wf = self._concus(self._wf, ac._wf, axis=0)
        self._F = self._concus(self._F, ac._F)
        m = self._concus(self.mag if self._have_mag() else None,
                         ac.mag if ac._have_mag() else None)
        p = self._concus(self.ph if self._have_ph() else None,
                         ac.ph if ac._have_ph() else None)
        self._mag, self._dmag = m, None
        self._ph, self._dph = p, None
        self._D = self._concus(self._D, ac._D)

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
        """util_audio.py:411-434 with the build-defined CQT (oracle/cqt.py);
        like the reference, `filter_scale` is ignored (:426) and only the
        magnitude is available (magnitude_only=False is not supported)."""
        if not magnitude_only:
            raise NotImplementedError('complex CQT output is not provided')
        if nbins is None:
            nbins = int((note_to_midi(highest_note) - note_to_midi(lowest_note)) * bins_per_tone)
        fmin = float(midi_to_hz(note_to_midi(lowest_note)))
        dev = require_gpu()
        wf = to_dev(np.asarray(self.wf, dtype=np.float32))[None]
        t = self._seconds_to_frames(start + duration)
        s = self._seconds_to_frames(start)
        Ttot = self.shape[1]
        s_c = max(0, min(s, Ttot))
        t_c = max(s_c, min(t, Ttot))
        rel = resize_source_frames(t_c - s_c, target_frame_count)
        src = np.where(rel < 0, -1, rel + s_c).astype(np.int32)
        table = cqt_table(self.sr, fmin, nbins, int(12 * bins_per_tone), dev)
        out = np.zeros((nbins, target_frame_count), dtype=np.float32)
        for c0 in range(0, target_frame_count, 8):
            cols = src[c0:c0 + 8]
            o = cqt_slices(wf, to_dev(cols[None], torch.int32), table, nbins, self.hl)
            out[:, c0:c0 + len(cols)] = o[0].cpu().numpy()
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

    def resize(self, start, duration, target_frame_count, attribs=['F']):
        """util_audio.py:469-507."""
        nac = audio_complete(None, self.N, hop_length=self.hl, center=self.center,
                             sample_rate=self.sr)
        if self._ref_mag is not None:
            nac._ref_mag = self._ref_mag
        t = self._seconds_to_frames(start + duration)
        s = self._seconds_to_frames(start)
        for attrib in attribs:
            if attrib == 'F':
                nac.F = self._resize(self.F[:, s:t], target_frame_count)
            elif attrib == 'mag':
                nac.mag = self._resize(self.mag[:, s:t], target_frame_count)
            elif attrib == 'ph':
                nac.ph = self._resize(self.ph[:, s:t], target_frame_count)
            elif attrib == 'D':
                nac.D = self._resize(self.D[:, s:t], target_frame_count)
            else:
                raise ValueError('Invalid attribute requested')
        return nac

    def spectral_flatness(self):
        raise NotImplementedError('spectral_flatness is a render sanity check outside the hot path')

    def save(self, filename, flac=True):
        """util_audio.py:520-527: the waveform as PCM-24 FLAC (the WAV branch is not provided)."""
        if not flac:
            raise NotImplementedError('only FLAC output is provided')
        from . import flac as _flac
        _flac.save_float(np.asarray(self.wf), filename, sr=self.sr, bps=24)


def _amplitude_to_db(S, ref, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db (plots only; util_audio.py:179)."""
    magnitude = np.abs(np.asarray(S))
    log_spec = 20.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 20.0 * np.log10(np.maximum(amin, np.abs(ref)))
    return np.maximum(log_spec, log_spec.max() - top_db)


def _db_to_amplitude(S_db, ref):
    return ref * np.power(10.0, 0.05 * np.asarray(S_db))
