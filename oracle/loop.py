"""Oracle (test infrastructure): CPU restatement of the detect -> subtract loop.

Per window it performs, with numpy only, the steps training.py:296-449 performs
on the residual window -- feature recipe (:333-388), classify() of each head
(predict branch), subtract (:449) -- using the *predicted* note, exactly as the
product's batched driver composes them (amt_saga/loop.py).  It is the checker
for the loop parity tests and the timed ``cpu_baseline`` ("port") of bench.py.

Inputs are plain arrays: the windows, the heads' weight dicts and configs, the
guess bank waveforms, the song-level normalisers.  Nothing here imports the
product package.
"""
import numpy as np

from . import audio as oa
from . import cqt as ocqt
from . import rdcnn as orc

TAIL_SECONDS = 1.0


def rint_clamp(x, lo, hi):
    v = np.rint(np.float32(x))
    if not (v == v):
        return lo
    return int(min(max(v, lo), hi))


class LoopOracle:
    def __init__(self, params, heads, weights, iters=1, subtract=True, prog_group=None,
                 bank_waves=None, dtype=np.float32, guess_fn=None):
        """weights: dict head-name -> weight dict ('timing_start', 'timing_end',
        'pitch', 'instrument', 'velocity'); bank_waves [G*n_pitch, Lg] float.
        guess_fn(program, pitch, velocity, frames) -> waveform: the product's 'render'
        guess mode (one rendered guess per decision, training.py:421-431) instead of the bank."""
        self.p = params
        self.heads = tuple(heads)
        self.w = weights
        self.iters = iters
        self.do_subtract = subtract
        self.dtype = dtype
        self.guess_fn = guess_fn
        self.cfg = {k: orc.head_config(params, k) for k in
                    ('timing', 'pitch', 'instrument', 'velocity')}
        self.prog_group = prog_group if prog_group is not None else np.zeros(params.instrument_classes, np.int32)
        p = params
        f_lo = float(oa.midi_to_hz(p.pitch_low))
        self.tab_pitch = ocqt.cqt_table(p.sr, f_lo, p.pitch_bands, 12 * p.pitch_bins_per_tone)
        self.tab_inst = ocqt.cqt_table(p.sr, f_lo, p.instrument_bands, 12 * p.instrument_bins_per_tone)
        self.vel_bpt = 2
        n_vel = self.vel_bpt * (p.pitch_high - p.pitch_low) + p.bins_velocity
        self.tab_vel = ocqt.cqt_table(p.sr, float(oa.midi_to_hz(p.pitch_low - 10)), n_vel, 12 * self.vel_bpt)
        if bank_waves is not None:
            self.bank_mag = [oa.magphase(oa.stft(w, p.N, p.H))[0] for w in np.asarray(bank_waves)]
            self.bank_max = [m.max() for m in self.bank_mag]
            self.bank_frames = self.bank_mag[0].shape[1]
        self.tail_frames = int(TAIL_SECONDS * p.sr / p.H)
        self.margins = []        # |frac - 0.5| of every rounded decision (near-tie reporting)
        # per-decision record of the last run_window(): (name, iteration, float value(s), margin, decided, forced)
        self.decisions = []
        self._force = None       # see run_window(force=...)
        self._it = 0

    def _predict(self, name, cfgname, x):
        y = orc.forward(self.w[name], self.cfg[cfgname], [np.asarray(x, np.float32)[None, :, :, None]],
                        self.dtype)
        return y[0]

    def _round(self, y, lo, hi, name='', col=None):
        margin = abs(float(y) - np.floor(float(y)) - 0.5)
        self.margins.append(margin)
        v = rint_clamp(y, lo, hi)
        forced = False
        if self._force is not None and col is not None:
            # teacher forcing at a near-tie: when this float sits within `band` of a rounding boundary the two
            # arithmetics may legitimately land on different sides; the run then continues with the decision
            # handed in (the product's), provided it is the integer on the other side of that boundary -- so
            # every later iteration of the window is still compared like any other
            ev, band = self._force
            g = int(ev[self._it][col])
            if g != v and margin < band.get(name, 0.0) and abs(g - v) == 1:
                v, forced = g, True
        self.decisions.append((name, self._it, float(y), margin, v, forced))
        return v

    def _argmax(self, pr, name='instrument', col=3):
        pr = np.asarray(pr)
        order = np.argsort(-pr, kind='stable')               # first maximum first, as np.argmax
        v, runner_up = int(order[0]), int(order[1]) if len(pr) > 1 else int(order[0])
        margin = float(pr[v] - pr[runner_up])
        forced = False
        if self._force is not None:
            # teacher forcing at a near-tie of the two LARGEST probabilities only: the other implementation may name the
            # runner-up class when it lies within the band of the winner, and no other class
            ev, band = self._force
            g = int(ev[self._it][col])
            if g != v and g == runner_up and margin < band.get(name, 0.0):
                v, forced = g, True
        self.decisions.append((name, self._it, pr.copy(), margin, v, forced))
        return v

    def run_window(self, wave, refs, window_id=0, force=None):
        """wave float32 [L]; refs dict(ref_mag, ref_C_1, ref_C_inst, ref_C_foc).
        force: None, or (events [iters, 7] of another implementation, {head name: tie band in output units}):
        decisions whose float lies within the band of a rounding tie adopt that implementation's integer (see
        _round).  self.decisions then lists every decision with its float, margin and whether it was forced.
        Returns (events [iters, 7] int32, residual magnitude [F, T] float32)."""
        p = self.p
        self._force = force
        self.decisions = []
        ac = oa.AudioCompleteOracle(np.asarray(wave, np.float32), p.N, p.H)
        ac.mag                                        # STFT + magphase (training.py:269)
        T = ac.shape[1]
        events = np.full((self.iters, 7), -1, np.int32)
        for it in range(self.iters):
            self._it = it
            if 'timing' in self.heads:
                ct = oa.AudioCompleteOracle.compress_bands(ac.mag, bands=p.timing_bands)
                ct = oa.AudioCompleteOracle._resize(ct, p.timing_frames) / refs['ref_mag']
                onset = self._round(self._predict('timing_start', 'timing', ct)[0], 0, T - 1, 'timing_start', 5)
                end = self._round(self._predict('timing_end', 'timing', ct)[0], 0, T, 'timing_end', 6)
            else:
                onset, end = 0, p.pitch_frames
            src = ocqt.slice_C_frames(T, onset, end, p.pitch_frames)
            need_wave = any(h in self.heads for h in ('pitch', 'instrument', 'velocity'))
            wf = ac.wf if need_wave else None         # iSTFT(mag*ph) after the first subtract
            pitch, program, velocity = 60, -1, -1
            if 'pitch' in self.heads:
                cp = ocqt.cqt_frames(wf, src, self.tab_pitch[0], self.tab_pitch[1], p.H) / refs['ref_C_1']
                pitch = self._round(self._predict('pitch', 'pitch', cp)[0], p.pitch_low, p.pitch_high, 'pitch', 2)
            if 'instrument' in self.heads:
                ci = ocqt.cqt_frames(wf, src, self.tab_inst[0], self.tab_inst[1], p.H) / refs['ref_C_inst']
                program = self._argmax(self._predict('instrument', 'instrument', ci))
            if 'velocity' in self.heads:
                b0 = self.vel_bpt * (pitch - p.pitch_low)
                cv = ocqt.cqt_frames(wf, src, self.tab_vel[0][b0:b0 + p.bins_velocity],
                                     self.tab_vel[1][b0:b0 + p.bins_velocity], p.H) / refs['ref_C_foc']
                velocity = self._round(self._predict('velocity', 'velocity', cv)[0], 1, 127, 'velocity', 4)
            if self.do_subtract:
                n_pitch = p.pitch_high - p.pitch_low + 1
                pr = min(max(program, 0), p.instrument_classes - 1) if program >= 0 else 0
                g = int(self.prog_group[pr]) * n_pitch + min(max(pitch - p.pitch_low, 0), n_pitch - 1)
                if self.guess_fn is not None:
                    gw = np.asarray(self.guess_fn(pr, pitch, velocity, max(end - onset, 0)), np.float32)
                    gmag = oa.magphase(oa.stft(gw, p.N, p.H))[0]
                    gmax, gframes = gmag.max(), gmag.shape[1]
                else:
                    gmag, gmax, gframes = self.bank_mag[g], self.bank_max[g], self.bank_frames
                gf = min(max(end - onset, 0) + self.tail_frames, gframes)
                # audio_complete.subtract with a magnitude subtrahend (util_audio.py:240-259),
                # offset given directly in frames
                mag_sub = gmag[:, :gf].copy()
                mag_sub *= ac.ref_mag / gmax
                if mag_sub.shape[1] + onset > T:
                    mag_sub = mag_sub[:, :T - onset]
                m = ac.mag
                m[:, onset:onset + mag_sub.shape[1]] -= mag_sub
                ac.mag = np.maximum(m, 0, m)
            events[it] = (window_id, it, pitch, program, velocity, onset, end)
        return events, ac.mag

    def ref_levels(self, wave):
        """Song-level constants (training.py:269-282, with one window standing for the song): ref_mag =
        max |STFT|; ref_C_* = max of the whole CQT -- every bin of the normaliser's grid, every frame."""
        p = self.p
        mag = oa.magphase(oa.stft(np.asarray(wave, np.float32), p.N, p.H))[0]
        f_lo = float(oa.midi_to_hz(p.pitch_low))
        span = p.pitch_high - p.pitch_low
        out = {'ref_mag': mag.max()}
        if 'pitch' in self.heads:
            t = ocqt.cqt_table(p.sr, f_lo, span, 12)
            out['ref_C_1'] = np.float32(ocqt.cqt_window_max(wave, t[0], t[1], p.H))
        if 'instrument' in self.heads:
            t = ocqt.cqt_table(p.sr, f_lo, span * p.instrument_bins_per_tone, 12 * p.instrument_bins_per_tone)
            out['ref_C_inst'] = np.float32(ocqt.cqt_window_max(wave, t[0], t[1], p.H))
        if 'velocity' in self.heads:
            t = ocqt.cqt_table(p.sr, f_lo, span * p.instrument_bins_per_tone * 4,
                               12 * p.instrument_bins_per_tone * 4)
            out['ref_C_foc'] = np.float32(ocqt.cqt_window_max(wave, t[0], t[1], p.H))
        return out
