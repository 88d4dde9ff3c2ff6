"""Oracle (test infrastructure): numpy restatement of util_audio.audio_complete.

Follows /root/reference/util_audio.py:32-527.  The librosa calls the reference
makes (librosa is NOT vendored and NOT installed; version unpinned, API usage
implies 0.6.x-0.7.x) are restated from librosa's documented behaviour:

    librosa.stft(y, n_fft, hop_length, center=True)   util_audio.py:127
        reflect-pad n_fft//2 both sides, frame at t*hop, periodic Hann(n_fft),
        rfft, complex64 result of shape (1+n_fft//2, 1+len(y)//hop)
    librosa.istft(F, hop_length, center=True)          util_audio.py:92-104
        irfft, periodic Hann synthesis window, overlap-add, divide by the
        window sum-of-squares where it exceeds tiny, trim n_fft//2 both ends,
        float32 result of length hop*(T-1)
    librosa.core.magphase(F)                           util_audio.py:147,162
        mag = |F| ; phase = exp(1j*angle(F))  (phase = 1 where F == 0)
    librosa.amplitude_to_db(S, ref)                    util_audio.py:179
        20*log10(max(1e-5,|S|)) - 20*log10(max(1e-5,ref)), floored at max-80 dB
    librosa.db_to_amplitude(D, ref)                    util_audio.py:101,124,145
        ref * 10**(D/20)
    librosa.core.fft_frequencies(sr, n_fft)            util_audio.py:67
    librosa.core.midi_to_hz(m)                         util_audio.py:281

This chain is pinned against the reference's recorded librosa output
(subtraction_demo/*.flac) by tests/test_oracle_golden.py.
"""
import bisect
import copy

import numpy as np


# ----------------------------------------------------------------------------
# librosa primitives (restated)
# ----------------------------------------------------------------------------
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) in float64."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def fft_frequencies(sr, n_fft):
    return np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)


def midi_to_hz(m):
    return 440.0 * (2.0 ** ((np.asanyarray(m, dtype=np.float64) - 69.0) / 12.0))


def reflect_index(i, n):
    """numpy.pad(mode='reflect') source index for padded position i (may be
    outside [0,n)); single reflection is enough for pad <= n-1."""
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    i = np.where(i >= n, 2 * (n - 1) - i, i)
    return i


def stft(y, n_fft, hop_length=None, center=True):
    """util_audio.py:127 -> librosa.stft. Returns complex64 [1+n_fft//2, T]."""
    y = np.asarray(y)
    hop = int(n_fft // 4) if hop_length is None else int(hop_length)
    if center:
        yp = np.pad(y.astype(np.float64), int(n_fft // 2), mode='reflect')
    else:
        yp = y.astype(np.float64)
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = yp[idx] * hann_periodic(n_fft)[:, None]
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def istft(F, hop_length=None, center=True):
    """util_audio.py:92-104 -> librosa.istft. Returns float32 [hop*(T-1)]."""
    F = np.asarray(F)
    n_fft = 2 * (F.shape[0] - 1)
    hop = int(n_fft // 4) if hop_length is None else int(hop_length)
    n_frames = F.shape[1]
    win = hann_periodic(n_fft)
    expected = n_fft + hop * (n_frames - 1)
    y = np.zeros(expected, dtype=np.float64)
    ytmp = win[:, None] * np.fft.irfft(F.astype(np.complex128), n=n_fft, axis=0)
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += ytmp[:, t]
    wss = np.zeros(expected, dtype=np.float64)
    w2 = win ** 2
    for t in range(n_frames):
        wss[t * hop:t * hop + n_fft] += w2
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    if center:
        y = y[int(n_fft // 2):-int(n_fft // 2)]
    return y.astype(np.float32)


def magphase(F):
    mag = np.abs(F)
    phase = np.exp(1.j * np.angle(F))
    return mag, phase.astype(np.complex64) if F.dtype == np.complex64 else phase


def amplitude_to_db(S, ref=1.0, amin=1e-5, top_db=80.0):
    magnitude = np.abs(np.asarray(S))
    ref_value = np.abs(ref)
    log_spec = 20.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 20.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def db_to_amplitude(S_db, ref=1.0):
    return ref * np.power(10.0, 0.05 * np.asarray(S_db))


def spectral_flatness(S, amin=1e-10, power=2.0):
    """librosa.feature.spectral_flatness(S=S): [1, T]."""
    S_thresh = np.maximum(amin, np.asarray(S, dtype=np.float64) ** power)
    gmean = np.exp(np.mean(np.log(S_thresh), axis=0, keepdims=True))
    amean = np.mean(S_thresh, axis=0, keepdims=True)
    return gmean / amean


# ----------------------------------------------------------------------------
# audio_complete restated  (util_audio.py:32-527)
# ----------------------------------------------------------------------------
class AudioCompleteOracle:
    """Lazy, mutually-invalidating cache wf <-> F <-> mag/ph <-> D.
    Layout is the reference's: spectra are [F, T] (bins x frames)."""

    def __init__(self, waveform, n_fft, hop_length=None, center=True,
                 sample_rate=44100):                       # util_audio.py:33-67
        self._wf = waveform
        self._F = None
        self._mag = None
        self._ref_mag = None
        self._ph = None
        self._D = None
        self.sr = sample_rate
        self.N = n_fft
        self.center = center
        self.hl = hop_length if hop_length is not None else int(np.floor(n_fft / 4))
        self._fft_freq = fft_frequencies(sample_rate, n_fft)

    def clone(self):                                       # util_audio.py:69-87
        ac = AudioCompleteOracle(copy.deepcopy(self._wf), self.N, self.hl,
                                 self.center, self.sr)
        ac._F = copy.deepcopy(self._F)
        ac._mag = copy.deepcopy(self._mag)
        ac._ref_mag = self._ref_mag
        ac._ph = copy.deepcopy(self._ph)
        ac._D = copy.deepcopy(self._D)
        return ac

    # -- properties ---------------------------------------------------------
    @property
    def wf(self):                                          # util_audio.py:88-106
        if self._wf is None:
            if self._F is not None:
                self._wf = istft(self._F, self.hl, self.center)
            elif self._mag is not None and self._ph is not None:
                self._F = self._mag * self._ph
                self._wf = istft(self._F, self.hl, self.center)
            elif self._D is not None and self._ph is not None:
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self._mag = db_to_amplitude(self._D, ref=self._ref_mag)
                self._F = self._mag * self._ph
                self._wf = istft(self._F, self.hl, self.center)
        return self._wf

    @wf.setter
    def wf(self, value):                                   # util_audio.py:107-114
        self._D = None
        self._ref_mag = None
        self._mag = None
        self._ph = None
        self._F = None
        self._wf = value

    @property
    def F(self):                                           # util_audio.py:116-129
        if self._F is None:
            if self._mag is not None and self._ph is not None:
                self._F = self._mag * self._ph
            elif self._D is not None and self._ph is not None:
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self._mag = db_to_amplitude(self._D, ref=self._ref_mag)
                self._F = self._mag * self._ph
            elif self.wf is not None:
                self._F = stft(self.wf, self.N, self.hl, self.center)
        return self._F

    @F.setter
    def F(self, value):                                    # util_audio.py:130-137
        self._D = None
        self._ref_mag = None
        self._mag = None
        self._ph = None
        self._F = value
        self._wf = None

    @property
    def mag(self):                                         # util_audio.py:139-148
        if self._mag is None:
            if self._D is not None and self._ph is not None:
                if self._ref_mag is None:
                    self._ref_mag = 1.0
                self._mag = db_to_amplitude(self._D, ref=self._ref_mag)
            else:
                self._mag, self._ph = magphase(self.F)
        return self._mag

    @mag.setter
    def mag(self, val):                                    # util_audio.py:149-157
        self._D = None
        self._ref_mag = None
        self._mag = val
        if self._ph is not None and self._ph.shape != val.shape:
            self._ph = None
        self._F = None
        self._wf = None

    @property
    def ph(self):                                          # util_audio.py:159-163
        if self._ph is None:
            self._mag, self._ph = magphase(self.F)
        return self._ph

    @ph.setter
    def ph(self, val):                                     # util_audio.py:164-168
        self._ph = val
        self._F = None
        self._wf = None

    @property
    def ref_mag(self):                                     # util_audio.py:170-174
        if self._ref_mag is None:
            self._ref_mag = np.max(self.mag)
        return self._ref_mag

    @property
    def D(self):                                           # util_audio.py:176-180
        if self._D is None:
            self._D = amplitude_to_db(self.mag, ref=self.ref_mag)
        return self._D

    @D.setter
    def D(self, val):                                      # util_audio.py:181-190
        self._D = val
        if self._ph is not None and self._ph.shape != val.shape:
            self._ph = None
        self._mag = None
        self._F = None
        self._wf = None

    def _P(self, name):                                    # util_audio.py:192-207
        if name == 'wf':
            return self._wf
        elif name == 'F':
            return self._F
        elif name == 'mag':
            return self._mag
        elif name == 'ph':
            return self._ph
        elif name == 'D':
            return self._D
        raise ValueError('Requested attribute does not exist')

    @property
    def shape(self):                                       # util_audio.py:209-218
        if self._mag is not None:
            return self._mag.shape
        if self._ph is not None:
            return self._ph.shape   # reference returns _mag.shape here (bug, :214)
        if self._D is not None:
            return self._D.shape
        return self.F.shape

    # -- the subtraction step ----------------------------------------------
    def subtract(self, subtrahend, offset=0, attack_compensation=0,
                 normalize=True, relu=True, overkill_factor=1):
        """util_audio.py:221-259."""
        if isinstance(subtrahend, type(self)):
            mag_sub = copy.deepcopy(subtrahend.mag)
            if normalize:
                mag_sub *= self.ref_mag / subtrahend.ref_mag
        else:
            mag_sub = copy.deepcopy(subtrahend)
            ref_max_sub = np.max(mag_sub)
            if normalize:
                mag_sub *= self.ref_mag / ref_max_sub
        mag_sub *= overkill_factor
        offset = max(self._seconds_to_frames(offset) - attack_compensation, 0)
        if mag_sub.shape[1] + offset > self.mag.shape[1]:
            mag_sub = mag_sub[:, :(self.mag.shape[1] - offset)]
        self.mag -= np.concatenate(
            (np.zeros((self.mag.shape[0], offset)),
             mag_sub,
             np.zeros((self.mag.shape[0],
                       self.mag.shape[1] - offset - mag_sub.shape[1]))), axis=1)
        if relu:
            self.mag = np.maximum(self.mag, 0, self.mag)

    def _seconds_to_frames(self, time):                    # util_audio.py:261-264
        return int(np.floor(time * self.shape[1] * self.sr / self.wf.shape[0]))

    def _frames_to_seconds(self, frames):                  # util_audio.py:269-272
        return frames / self.shape[1] / self.sr * self.wf.shape[0]

    def midi_tone_to_FFT(self, tone):                      # util_audio.py:278-284
        f = midi_to_hz(tone)
        ind = bisect.bisect_right(self._fft_freq, f) - 1
        ind = 0 if ind == 0 else ind - 1
        return ind

    def section(self, start, end, duration_in_frames=None):  # util_audio.py:286-328
        tfs = self._seconds_to_frames(start)
        if duration_in_frames is None:
            tfe = self._seconds_to_frames(end)
        else:
            tfe = tfs + duration_in_frames
        if self._wf is not None:
            wav_start = int(np.floor(self._frames_to_seconds(tfs) * self.sr))
            wav_end = int(np.floor(self._frames_to_seconds(tfe) * self.sr))
            wav_cp = copy.deepcopy(self.wf[wav_start:wav_end])
            if wav_cp.shape[0] < wav_end - wav_start:
                wav_cp = np.concatenate((wav_cp, np.zeros(wav_end - wav_cp.shape[0])))
        else:
            wav_cp = None
        nac = AudioCompleteOracle(wav_cp, self.N, self.hl, self.center, self.sr)

        def cc(f):
            if f is not None:
                cpd = copy.deepcopy(f[:, tfs:tfe])
                if f.shape[1] >= tfe:
                    return cpd
                return np.concatenate(
                    (cpd, np.zeros((f.shape[0], tfe - f.shape[1]))), axis=1)
        nac._F = cc(self._F)
        nac._ref_mag = self._ref_mag
        nac._mag = cc(self._mag)
        nac._ph = cc(self._ph)
        nac._D = cc(self._D)
        return nac

    def spectral_flatness(self):                           # util_audio.py:330-332
        S = np.abs(stft(self.wf, self.N, self.hl))
        return np.mean(spectral_flatness(S))

    def section_power(self, name, band_min, band_max):     # util_audio.py:334-349
        P = self._P(name)
        h = P.shape[0]
        cpd = copy.deepcopy(P[band_min:band_max, :])
        if band_max > h:
            cpd = np.concatenate((cpd, np.zeros((band_max - h, P.shape[1]))), axis=0)
        return cpd

    def slice(self, start_in_frames, end_in_frames):       # util_audio.py:351-365
        if self._wf is not None:
            self._wf = self._wf[int(self._frames_to_seconds(start_in_frames) * self.sr):
                                int(self._frames_to_seconds(end_in_frames) * self.sr)]
        if self._F is not None:
            self._F = self._F[:, start_in_frames:end_in_frames]
        if self._mag is not None:
            self._mag = self._mag[:, start_in_frames:end_in_frames]
        if self._ph is not None:
            self._ph = self._ph[:, start_in_frames:end_in_frames]
        if self._D is not None:
            self._D = self._D[:, start_in_frames:end_in_frames]

    @staticmethod
    def _concus(dest, src, axis=1):                        # util_audio.py:368-372
        if src is None or dest is None:
            return None
        return np.concatenate((dest, src), axis=axis)

    def concat(self, ac):                                  # util_audio.py:374-382
        self._wf = self._concus(self._wf, ac._wf, axis=0)
        self._F = self._concus(self._F, ac._F)
        self._mag = self._concus(self._mag, ac._mag)
        self._ph = self._concus(self._ph, ac._ph)
        self._D = self._concus(self._D, ac._D)

    @staticmethod
    def _resize(P, target_frame_count):                    # util_audio.py:384-409
        t = P.shape[1]
        if t == 0:
            return np.zeros((P.shape[0], target_frame_count))
        elif t == target_frame_count:
            resd = P
        elif t < 3:
            resd = np.concatenate(
                (P[:, :1], np.tile(P[:, -1:], target_frame_count - 1)), axis=1)
        elif t < target_frame_count:
            lim = np.min((1, int(np.round(t / 3))))
            l_t = int(np.floor((target_frame_count - 2 * lim) / (t - 2 * lim)))
            tiled = np.tile(P[:, lim:-lim], l_t)
            resd = np.concatenate(
                (P[:, :lim], tiled,
                 P[:, -(target_frame_count - tiled.shape[1] - lim):]), axis=1)
        else:
            resd = P[:, :target_frame_count]
        return resd

    @staticmethod
    def compress_bands(spectrum, bands=80, log=True):      # util_audio.py:436-466
        ns = np.zeros((bands, spectrum.shape[1]))
        if log:
            ind = band_edges(spectrum.shape[0], bands)
            for i in range(bands):
                ns[i, :] = np.mean(spectrum[int(ind[i]):int(ind[i + 1]), :], axis=0)
        else:
            r = spectrum.shape[0] // bands
            for i in range(bands):
                ns[i, :] = np.mean(spectrum[r * i:r * (i + 1), :], axis=0)
        return ns

    def resize(self, start, duration, target_frame_count, attribs=['F']):
        """util_audio.py:469-507 (the 'D' branch's ph mix-up, :503, is NOT
        replicated; it is off the hot path)."""
        nac = AudioCompleteOracle(None, self.N, self.hl, self.center, self.sr)
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


def band_edges(n_rows, bands):
    """util_audio.py:451-456: geomspace(1, n_rows, bands+1) truncated to int,
    first edge forced to 0, each band widened to at least one row."""
    ind = np.geomspace(1, n_rows, bands + 1).astype(np.int64)   # dtype=np.int truncates
    ind[0] = 0
    for i in range(bands):
        sub = ind[i + 1] - ind[i]
        if sub < 1:
            ind[i + 1] += -sub + 1
    return ind


def resize_index_map(t, target):
    """Source-column index for every output column of _resize (util_audio.py:
    384-409) as an int array of length `target`; -1 means "zero column"
    (the t == 0 case).  Used by the tests to check the product's gather table
    against the reference's concatenate/tile formulation."""
    if t == 0:
        return -np.ones(target, dtype=np.int64)
    src = np.arange(t)[None, :]
    out = AudioCompleteOracle._resize(src, target)
    return np.asarray(out[0], dtype=np.int64)
