"""The detect -> subtract loop for B windows at once, device resident.

The reference has no inference loop (main.py only trains); what it
has is the per-note generator loop of training.py:296-449 that builds the head
inputs from the residual window and subtracts the *gold* note.  This driver
composes the same steps with the *predicted* note, for B independent windows
per launch:

  per iteration (one note per window):
    C_timing  = compress_bands(mag, 20) / ref_mag_song        training.py:333-336
    onset,end = rint(timing_start(C_timing)), rint(timing_end(C_timing))
    wf        = istft(mag * ph)                               util_audio.py:94-97 (what slice_C reads)
    C_sw_pitch = slice_C(onset, dur, 8, bpt=2) / ref_C_1      training.py:340-342
    pitch     = rint(pitch_classifier(C_sw_pitch))
    C_sw_inst = slice_C(onset, dur, 8, bpt=4) / ref_C_inst    training.py:343-346
    program   = argmax(InstrumentClassifier(C_sw_inst))
    C_velocity = slice_C(..., bpt=2, nbins=36, lowest=pitch-10) / ref_C_foc   training.py:382-388
    velocity  = rint(VelocityClassifier(C_velocity))
    guess     = template bank[(program group, pitch)]         stand-in for render(), synth.py
                (guess='render': one note of the decided duration synthesised per window and
                 iteration by amt_synth_windows and STFT'd -- what the reference does per note,
                 training.py:421-431 -- instead of the fixed-duration bank row)
    mag       = relu(mag - guess * ref_mag/ref_mag(guess)) at frame `onset`   training.py:449

Everything numeric is a HIP kernel behind the C ABI; torch only holds the
buffers.  Song-level constants (ref_mag_song, ref_C_*) are computed once in
prepare(), as the reference computes them once per song (training.py:269-282).
"""
import ctypes as C

import os

import numpy as np
import torch

from . import _lib, synth
from .audio import AudioBatch, cqt_slices, cqt_table, cqt_window_max, midi_to_hz
from .device import empty, ptr, require_gpu, stream_ptr, to_dev, zeros
from .heads import (InstrumentClassifier, VelocityClassifier, pitch_classifier,
                    timming_classifier)

EVENT_FIELDS = ('window', 'iter', 'pitch', 'program', 'velocity', 'onset_frame', 'end_frame')


class TranscriptionLoop:
    def __init__(self, params, heads=('timing', 'pitch', 'velocity'), iters=1, subtract=True,
                 groups=(0,), seeds=None, guess='bank', timbres=None, soundfont=None):
        self.p = params
        self.heads = tuple(heads)
        self.iters = int(iters)
        self.do_subtract = bool(subtract)
        self.groups = tuple(groups)
        if guess not in ('bank', 'render'):
            raise ValueError('Requested attribute does not exist')
        self.guess = guess
        # timbres = 'gm' (guess='render' only): every decided MIDI program is synthesised with its own timbre
        # (synth.gm_timbre_table) instead of one of the three groups
        if timbres not in (None, 'gm'):
            raise ValueError("timbres: None or 'gm'")
        if timbres == 'gm' and guess != 'render':
            raise ValueError("per-program timbres need guess='render' (the template bank holds the three groups)")
        self.timbres = timbres
        # soundfont (guess='render' only): an sf2.SoundFont or a path -- the guess is played from its samples
        # (main.py:25-29 -soundfont_path; util_audio.py:758-786), every decided MIDI program through its own preset
        if soundfont is not None and guess != 'render':
            raise ValueError("a soundfont needs guess='render'")
        if soundfont is not None and not hasattr(soundfont, 'programs'):
            from . import sf2 as _sf2
            soundfont = _sf2.SoundFont(soundfont)
        self.soundfont = soundfont
        # AMT_TIMING_STREAMS=2: timing_end on a second HIP stream under timing_start (-1 ... +3 % on a C3 step, run to run).
        # Opt-in.  Until round 4's fix two timing networks that really overlapped in time returned wrong floats at the
        # metric size: packed-FP32 vector instructions (v_pk_mul / add / fma_f32) of the transform kernels compute wrong
        # values in lanes 48-63 while a wave of ANOTHER dispatch issues v_mfma_f32_16x16x32_f16 on the same SIMD
        # (DESIGN 10.1); the network kernels are built without those instructions now (build.py NO_PK) and the full-size
        # fixtures pass on two streams (tests/test_gpu_fullsize_fixtures.py)
        self.timing_streams = int(os.environ.get('AMT_TIMING_STREAMS', '1'))
        # the subtraction on the guess's frames only (amt_subtract_span): the residual is a magnitude spectrogram (>= 0) and
        # the timing features' compress_bands pass leaves the per-frame maxima on the way; AMT_SUBTRACT_SPAN=0: whole windows
        self.span_subtract = os.environ.get('AMT_SUBTRACT_SPAN', '1') != '0'
        # diagnostic hook: when set to a list, iterate() appends one dict per iteration with copies of the heads'
        # pre-rounding outputs (what res_net.predict returns, RDCNN.py:591-597) -- the parity tests compare them
        # with the oracle's floats so that no window near a rounding tie leaves a test uncompared
        self.trace = None
        self.lib = _lib.load()
        # default seeds: synthetic timing_start / timing_end nets whose (nearly input-independent)
        # outputs satisfy onset < end, so the short-window features are not empty
        seeds = seeds or {}
        self.nets = {}
        if 'timing' in self.heads:
            self.nets['timing_start'] = timming_classifier(params, weight_seed=seeds.get('timing_start', 107))
            self.nets['timing_end'] = timming_classifier(params, weight_seed=seeds.get('timing_end', 105))
        if 'pitch' in self.heads:
            self.nets['pitch'] = pitch_classifier(params, weight_seed=seeds.get('pitch', 101))
        if 'instrument' in self.heads:
            self.nets['instrument'] = InstrumentClassifier(params, 'instrument',
                                                           weight_seed=seeds.get('instrument', 102))
        if 'velocity' in self.heads:
            self.nets['velocity'] = VelocityClassifier(params, weight_seed=seeds.get('velocity', 103))
        self._dev_ready = False

    # ---- one-time device setup (not timed: weights / tables / bank upload) ---------
    def setup_device(self, bank_waves=None):
        p = self.p
        dev = require_gpu()
        sr, lo = p.sr, p.pitch_low
        f_lo = float(midi_to_hz(lo))
        self.tab_pitch = cqt_table(sr, f_lo, p.pitch_bands, 12 * p.pitch_bins_per_tone, dev)
        self.tab_inst = cqt_table(sr, f_lo, p.instrument_bands, 12 * p.instrument_bins_per_tone, dev)
        # velocity: 36 bins at 2 bins/semitone from (pitch-10) -> one global grid from midi lo-10
        self.vel_bpt = 2
        n_vel = self.vel_bpt * (p.pitch_high - lo) + p.bins_velocity
        self.tab_vel = cqt_table(sr, float(midi_to_hz(lo - 10)), n_vel, 12 * self.vel_bpt, dev)
        # song-level normaliser grids (training.py:271-282): bpt 1, inst_bpt, 4*inst_bpt over A0..C8
        span = p.pitch_high - lo
        self.tab_ref1 = cqt_table(sr, f_lo, span * 1, 12, dev)
        self.tab_refi = cqt_table(sr, f_lo, span * p.instrument_bins_per_tone,
                                  12 * p.instrument_bins_per_tone, dev)
        self.tab_reff = cqt_table(sr, f_lo, span * p.instrument_bins_per_tone * 4,
                                  12 * p.instrument_bins_per_tone * 4, dev)
        # index of a program group inside this loop's bank
        remap = np.zeros(3, dtype=np.int32)
        for i, g in enumerate(self.groups):
            remap[g] = i
        self.prog_group = to_dev(remap[synth.prog_group_table(p.instrument_classes)], torch.int32)
        self.prog_preset = to_dev(np.arange(p.instrument_classes, dtype=np.int32)
                                  if (self.timbres == 'gm' or self.soundfont is not None)
                                  else synth.prog_group_table(p.instrument_classes), torch.int32)
        self.bank_dur = 1.0
        self.bank_len = int(round((self.bank_dur + synth.TAIL_SECONDS) * sr))
        if bank_waves is None:
            bank_waves = synth.guess_bank_waves(self.groups, p.pitch_low, p.pitch_high, sr=sr, device=dev)
        bank = AudioBatch(bank_waves, p.N, p.H).stft(with_phase=False)
        self.bank_mag, self.bank_max, self.bank_frames = bank.mag, bank.ref_max, bank.T
        self.tail_frames = int(synth.TAIL_SECONDS * sr / p.H)
        for n in self.nets.values():
            n._ensure()
        self._dev_ready = True
        return self

    # ---- small wrappers over the glue kernels -----------------------------------------
    def _round(self, y, lo, hi):
        out = empty((y.shape[0],), torch.int32)
        _lib.check(self.lib.amt_round_clamp(ptr(y), y.shape[0], y.stride(0) if y.dim() > 1 else 1,
                                            int(lo), int(hi), ptr(out), stream_ptr()))
        return out

    def _argmax(self, pr):
        out = empty((pr.shape[0],), torch.int32)
        _lib.check(self.lib.amt_argmax_rows(ptr(pr), pr.shape[0], pr.shape[1], ptr(out), stream_ptr()))
        return out

    def _resize_table(self, s, e, T, frames):
        out = empty((s.shape[0], frames), torch.int32)
        _lib.check(self.lib.amt_resize_table(ptr(s), ptr(e), s.shape[0], int(T), int(frames), ptr(out),
                                             stream_ptr()))
        return out

    # ---- per batch ------------------------------------------------------------------------
    def prepare(self, wave, refs=None):
        """STFT of the windows + the song-level constants (training.py:269-282; each window stands for its
        song): ref_mag = max |STFT|, ref_C_* = max over every bin and every frame of the normaliser's CQT
        grid.  `refs` may supply dict(ref_mag, ref_C_1, ref_C_inst, ref_C_foc) tensors [B]."""
        if not self._dev_ready:
            self.setup_device()
        p = self.p
        # the unit phase is read only by the iSTFT that feeds the CQT heads from iteration 1 on (iteration 0
        # reads the original samples): with one iteration, or without those heads, it is never stored
        # (8 F T bytes per window, 57 % of the STFT's traffic)
        odd_length = wave.shape[-1] % p.H != 0              # then even iteration 0 resynthesises (iterate())
        b = AudioBatch(wave, p.N, p.H).stft(with_phase=self.needs_phase or (self.needs_wave and odd_length))
        refs = dict(refs or {})
        if 'ref_mag' not in refs:
            refs['ref_mag'] = b.ref_max.clone()
        need_cqt = any(h in self.heads for h in ('pitch', 'instrument', 'velocity'))
        if need_cqt:
            if 'pitch' in self.heads and 'ref_C_1' not in refs:
                refs['ref_C_1'] = cqt_window_max(b.wave, self.tab_ref1, p.H)
            if 'instrument' in self.heads and 'ref_C_inst' not in refs:
                refs['ref_C_inst'] = cqt_window_max(b.wave, self.tab_refi, p.H)
            if 'velocity' in self.heads and 'ref_C_foc' not in refs:
                refs['ref_C_foc'] = cqt_window_max(b.wave, self.tab_reff, p.H)
        self.refs = refs
        return b

    def song_levels(self, song):
        """The song-level constants exactly as the reference computes them (training.py:269-282): ref_mag = the
        maximum of the whole song's |STFT|, ref_C_* = the maximum of the whole song's CQT on the respective grid.
        song: [n_samples] float32 device tensor.  Returns a dict of 0-d device tensors for the heads of this loop."""
        if not self._dev_ready:
            self.setup_device()
        p = self.p
        s = song.reshape(1, -1).to(torch.float32).contiguous()
        out = {'ref_mag': AudioBatch(s, p.N, p.H).stft(with_phase=False).ref_max[0].clone()}
        if 'pitch' in self.heads:
            out['ref_C_1'] = cqt_window_max(s, self.tab_ref1, p.H)[0]
        if 'instrument' in self.heads:
            out['ref_C_inst'] = cqt_window_max(s, self.tab_refi, p.H)[0]
        if 'velocity' in self.heads:
            out['ref_C_foc'] = cqt_window_max(s, self.tab_reff, p.H)[0]
        return out

    @property
    def needs_wave(self):
        return any(h in self.heads for h in ('pitch', 'instrument', 'velocity'))

    @property
    def needs_phase(self):
        return self.iters > 1 and self.needs_wave

    def iterate(self, b, it, events, window0=0):
        p = self.p
        B, T = b.mag.shape[0], b.mag.shape[1]
        st = stream_ptr()
        onset = end = pitch = program = velocity = None
        tr = {}
        if 'timing' in self.heads:
            ct = b.compress_bands(p.timing_bands, self.refs['ref_mag'], p.timing_frames, fmax=self.span_subtract)
            if self.timing_streams == 2:
                # the two timing networks read the same features and are independent: timing_end on a second stream
                # fills the tails of timing_start's small-image launches (5 x 8 and 10 x 64 layers)
                cur = torch.cuda.current_stream()
                if getattr(self, '_side_stream', None) is None:
                    self._side_stream = torch.cuda.Stream()
                side = self._side_stream
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    te = self.nets['timing_end'].classify(ct)
                ts = self.nets['timing_start'].classify(ct)
                cur.wait_stream(side)
                te.record_stream(cur)
            else:
                ts = self.nets['timing_start'].classify(ct)
                te = self.nets['timing_end'].classify(ct)
            onset = self._round(ts, 0, T - 1)
            end = self._round(te, 0, T)
            tr['timing_start'], tr['timing_end'] = ts, te
        else:
            onset = zeros((B,), torch.int32)
            end = torch.full((B,), p.pitch_frames, dtype=torch.int32, device=onset.device)
        src = self._resize_table(onset, end, T, p.pitch_frames)
        need_wave = any(h in self.heads for h in ('pitch', 'instrument', 'velocity'))
        wave_r = b.wave if (it == 0 and b.wave is not None and b.wave.shape[1] == p.H * (T - 1)) \
            else None
        if need_wave and wave_r is None:
            wave_r = b.istft()
        if 'pitch' in self.heads:
            cp = cqt_slices(wave_r, src, self.tab_pitch, p.pitch_bands, p.H, ref=self.refs['ref_C_1'])
            tr['pitch'] = self.nets['pitch'].classify(cp)
            pitch = self._round(tr['pitch'], p.pitch_low, p.pitch_high)
        else:
            pitch = torch.full((B,), 60, dtype=torch.int32, device=onset.device)
        if 'instrument' in self.heads:
            ci = cqt_slices(wave_r, src, self.tab_inst, p.instrument_bands, p.H, ref=self.refs['ref_C_inst'])
            tr['instrument'] = self.nets['instrument'].classify(ci)
            program = self._argmax(tr['instrument'])
        if 'velocity' in self.heads:
            bin0 = empty((B,), torch.int32)
            _lib.check(self.lib.amt_affine_i32(ptr(pitch), B, self.vel_bpt, -self.vel_bpt * p.pitch_low,
                                               ptr(bin0), st))
            cv = cqt_slices(wave_r, src, self.tab_vel, p.bins_velocity, p.H, bin0=bin0,
                            ref=self.refs['ref_C_foc'])
            tr['velocity'] = self.nets['velocity'].classify(cv)
            velocity = self._round(tr['velocity'], 1, 127)
        if self.do_subtract:
            gidx = empty((B,), torch.int32)
            gfr = empty((B,), torch.int32)
            _lib.check(self.lib.amt_note_select(
                ptr(program), ptr(pitch), ptr(onset), ptr(end), ptr(self.prog_group),
                self.prog_group.shape[0], B, p.pitch_low, p.pitch_high - p.pitch_low + 1,
                self.tail_frames, self.bank_frames, ptr(gidx), ptr(gfr), st))
            if self.guess == 'bank':
                b.subtract(self.bank_mag, self.bank_max, gidx, gfr, onset, normalize=True, relu=True,
                           span=self.span_subtract)
            else:
                notes = empty((B, 1, 5))
                _lib.check(self.lib.amt_guess_notes(
                    ptr(program), ptr(pitch), ptr(velocity), ptr(onset), ptr(end), ptr(self.prog_preset),
                    self.prog_preset.shape[0], B, p.H / p.sr, self.bank_dur, 100.0, ptr(notes), st))
                if self.soundfont is not None:
                    from . import sf2 as _sf2
                    gw = _sf2.render_windows_device(notes, self.bank_len, self.soundfont, p.sr)
                else:
                    gw = synth.render_windows_device(notes, self.bank_len, p.sr, timbres=self.timbres)
                g = AudioBatch(gw, p.N, p.H).stft(with_phase=False)
                b.subtract(g.mag, g.ref_max, None, gfr, onset, normalize=True, relu=True, span=self.span_subtract)
        if self.trace is not None:
            self.trace.append({k: v.clone() for k, v in tr.items()})
        _lib.check(self.lib.amt_pack_events(B, int(window0), int(it), ptr(pitch), ptr(program),
                                            ptr(velocity), ptr(onset), ptr(end), ptr(events[it]), st))

    def run(self, wave, window0=0, refs=None):
        """All iterations for one batch.  Returns (events [iters, B, 7] int32 device,
        the AudioBatch holding the residual)."""
        b = self.prepare(wave, refs)
        events = empty((self.iters, b.mag.shape[0], len(EVENT_FIELDS)), torch.int32)
        for it in range(self.iters):
            self.iterate(b, it, events, window0)
        return events, b

    def run_stream(self, host_batches, refs=None, window0=0):
        """run() over a sequence of HOST batches with the host -> HBM copy of batch i+1 overlapped with the
        compute of batch i: two device staging buffers, a copy stream, and events in both directions (the
        compute waits for its batch's copy; a copy waits until the compute that last read its buffer is done).
        host_batches: iterable of float32 [B, L] CPU tensors or arrays (pinned memory makes the copy
        asynchronous; pageable memory still works, serialised by the driver).  `refs`: dict of [B] device
        tensors shared by all batches, or a callable batch_index -> dict, or None (prepare() computes them).
        Yields (events, AudioBatch) per batch, in order; the AudioBatch's `wave` is the staging buffer and is
        overwritten two batches later."""
        if not self._dev_ready:
            self.setup_device()
        cur = torch.cuda.current_stream()
        if getattr(self, '_copy_stream', None) is None:
            self._copy_stream = torch.cuda.Stream()
        cs = self._copy_stream
        bufs = [None, None]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]

        def issue(i, host):
            h = host if torch.is_tensor(host) else torch.from_numpy(np.ascontiguousarray(host, dtype=np.float32))
            s = i & 1
            if bufs[s] is None or bufs[s].shape != h.shape:
                bufs[s] = empty(tuple(h.shape))
                cs.wait_stream(cur)                  # the allocation (and whatever freed that memory) is ordered
            with torch.cuda.stream(cs):
                cs.wait_event(free[s])               # no-op until the event has been recorded once
                bufs[s].copy_(h, non_blocking=True)
                ready[s].record(cs)

        it = iter(host_batches)
        nxt = next(it, None)
        if nxt is None:
            return
        issue(0, nxt)
        i = 0
        while nxt is not None:
            nxt = next(it, None)
            if nxt is not None:
                issue(i + 1, nxt)                    # in flight while batch i computes
            s = i & 1
            cur.wait_event(ready[s])
            r = refs(i) if callable(refs) else refs
            events, b = self.run(bufs[s], window0=window0, refs=r)
            free[s].record(cur)
            window0 += bufs[s].shape[0]
            yield events, b
            events = b = None                        # let the allocator reuse the batch's blocks
            i += 1

    def flops_per_window_iter(self):
        return sum(n.flops_per_window for n in self.nets.values())
