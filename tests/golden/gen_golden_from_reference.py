#!/usr/bin/env python3
"""Emit golden vectors by running the REFERENCE's own code (this container only).

Run:  python tests/golden/gen_golden_from_reference.py
Writes tests/golden/reference_vectors.npz (data only: inputs + expected outputs).

What is imported from /root/reference (read-only, no bytecode written):
  * util_train_test.py  -- pure numpy; needs the ``np.int = int`` shim on
    numpy >= 1.24 (util_train_test.py:29).
  * util_audio.py       -- its module-level imports of magenta / librosa /
    soundfile fail here (ordinary ModuleNotFoundError: the packages are not
    installed).  Empty placeholder modules satisfy the *import statements*;
    the only librosa functions the exercised code paths call are the two
    closed-form helpers ``fft_frequencies`` (util_audio.py:67) and
    ``midi_to_hz`` (util_audio.py:281), provided below.  Every other librosa
    entry point raises, so a vector can never silently contain anything but
    the reference's own numpy control flow:
        subtract, _seconds_to_frames, _frames_to_seconds, midi_tone_to_FFT,
        section, section_power, slice, concat, _resize, resize,
        compress_bands, property setters' invalidation.
    STFT / iSTFT / magphase are NOT exercised here (arrays are injected into
    ``_mag`` / ``_ph`` / ``_wf`` directly); that chain is pinned by the
    recorded FLAC triples instead (gen_golden_from_flac.py).

The reference never travels to the GPU box; only the .npz does.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_vectors.npz')


def _install_placeholders():
    if not hasattr(np, 'int'):
        np.int = int                                        # util_train_test.py:29

    class _Refuse(types.ModuleType):
        def __getattr__(self, name):
            raise RuntimeError('librosa.%s is not available (librosa absent); '
                               'golden vectors must not depend on it' % name)

    librosa = _Refuse('librosa')
    core = _Refuse('librosa.core')
    core.fft_frequencies = lambda sr, n_fft: np.linspace(
        0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    core.midi_to_hz = lambda notes: 440.0 * (2.0 ** ((np.asanyarray(notes) - 69.0) / 12.0))
    librosa.core = core
    sys.modules['librosa'] = librosa
    sys.modules['librosa.core'] = core
    for name in ('magenta', 'magenta.music', 'magenta.music.midi_io',
                 'magenta.protobuf', 'magenta.protobuf.music_pb2', 'soundfile'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['magenta.music'].midi_io = sys.modules['magenta.music.midi_io']
    sys.modules['magenta.protobuf'].music_pb2 = sys.modules['magenta.protobuf.music_pb2']
    import matplotlib
    matplotlib.use('Agg')


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    import util_train_test as utt
    import util_audio as ua

    g = {}
    rng = np.random.default_rng(20191003)

    # ---- a16 Hyperparams (util_train_test.py:15-79) -------------------------
    fields = ['N', 'sr', 'H', 'window_size_note_time', 'convolutional_layer_count',
              'pool_layer_frequency', 'feature_expand_frequency', 'timing_frames',
              'timing_bands', 'pitch_frames', 'pitch_low', 'pitch_high',
              'pitch_bins_per_tone', 'pitch_bands', 'instrument_frames',
              'instrument_bins_per_tone', 'instrument_bands', 'instrument_classes',
              'bins_velocity', 'velocity_min', 'velocity_max', 'batch_size']
    shapes = ['kernel_size_timing', 'pool_size_timing', 'kernel_size_pitch',
              'pool_size_pitch', 'kernel_size_instrument', 'pool_size_instrument',
              'kernel_size_velocity', 'pool_size_velocity',
              'residual_layer_frequencies']
    for tag, kw in (('hp4096', dict()), ('hp2048', dict(N=2048)),
                    ('hp2048_b2', dict(N=2048, bins_per_tone=2))):
        p = utt.Hyperparams('data', 'sf.sf2', **kw)
        g[tag + '_fields'] = np.array([float(getattr(p, f)) for f in fields])
        for s in shapes:
            g[tag + '_' + s] = np.array(getattr(p, s), dtype=np.int64)
    g['hp_field_names'] = np.array(fields)

    # ---- a17 check_shape / list_to_nd_array (util_train_test.py:93-146) -----
    specs = [rng.standard_normal((174, 8)).astype(np.float32) for _ in range(5)]
    labels = [60.0, 61.5, 21.0, 108.0, 77.0]
    ex, gold = utt.list_to_nd_array(specs, labels)
    g['l2nd_in'] = np.stack(specs)
    g['l2nd_labels'] = np.array(labels)
    g['l2nd_x'] = ex
    g['l2nd_y'] = gold
    dual = [(specs[i], specs[(i + 1) % 5]) for i in range(3)]
    exd, goldd = utt.list_to_nd_array(dual, labels[:3])
    g['l2nd_dual_x0'] = exd[0]
    g['l2nd_dual_x1'] = exd[1]
    g['l2nd_dual_y'] = goldd
    ex1, gold1 = utt.list_to_nd_array(specs[0], np.array([60.0]))
    g['l2nd_single_x'] = ex1
    g['l2nd_single_y'] = gold1
    ok = []
    for spec, b, f in ((specs, 174, 8), (specs[0], 174, 8), (dual, 174, 8),
                       (specs, 348, 8), (specs[0], 174, 9), (tuple(specs), 174, 8)):
        try:
            utt.check_shape(spec, b, f)
            ok.append(1)
        except ValueError as e:
            ok.append(0)
            g['check_shape_msg'] = np.array(str(e))
    g['check_shape_ok'] = np.array(ok)

    # ---- audio_complete pure-numpy methods ----------------------------------
    def make_ac(n_fft, T, hop=None, seed=0, with_ph=True):
        r = np.random.default_rng(seed)
        hop = n_fft // 4 if hop is None else hop
        Fb = n_fft // 2 + 1
        L = hop * (T - 1)
        ac = ua.audio_complete(np.zeros(L, dtype=np.float32), n_fft)
        ac._mag = (r.random((Fb, T)) ** 3).astype(np.float32)
        if with_ph:
            ang = r.uniform(-np.pi, np.pi, (Fb, T))
            ac._ph = np.exp(1j * ang).astype(np.complex64)
        return ac

    # a8 frame<->second maps and a12 midi_tone_to_FFT
    for n_fft, T in ((2048, 516), (4096, 258), (4096, 130)):
        ac = make_ac(n_fft, T, seed=1, with_ph=False)
        times = np.array([0.0, 0.01, 0.5, 0.4999, 1.0, 2.999, 3.0, 5.9, 6.5])
        g['s2f_%d_%d_t' % (n_fft, T)] = times
        g['s2f_%d_%d' % (n_fft, T)] = np.array([ac._seconds_to_frames(t) for t in times])
        fr = np.array([0, 1, 21, 43, 129, T - 1, T])
        g['f2s_%d_%d_f' % (n_fft, T)] = fr
        g['f2s_%d_%d' % (n_fft, T)] = np.array([ac._frames_to_seconds(f) for f in fr])
        g['tone2fft_%d' % n_fft] = np.array([ac.midi_tone_to_FFT(m) for m in range(0, 128)])

    # a7 subtract: flag combinations, offsets incl. overrun clipping
    case = 0
    for n_fft, T, Tg in ((256, 40, 13), (512, 24, 9)):
        for (offset_s, ac_comp, normalize, relu, overkill) in (
                (0.0, 0, True, True, 1), (0.1, 0, True, True, 1),
                (0.1, 2, True, True, 1), (0.1, 50, True, True, 1),
                (0.2, 0, False, True, 1), (0.2, 0, True, False, 1),
                (0.2, 1, True, True, 1.5), (0.35, 0, True, True, 1),
                (0.4, 0, False, False, 0.5), (0.0, 0, True, True, 2)):
            mix = make_ac(n_fft, T, seed=100 + case)
            gs = make_ac(n_fft, Tg, seed=200 + case)
            # offsets above are fractions of the window; convert to seconds
            offset_s = offset_s * len(mix._wf) / 44100.0 * 2.0
            g['sub%d_mix' % case] = mix._mag.copy()
            g['sub%d_guess' % case] = gs._mag.copy()
            g['sub%d_args' % case] = np.array(
                [n_fft, offset_s, ac_comp, int(normalize), int(relu), overkill],
                dtype=np.float64)
            mix.subtract(gs, offset=offset_s, attack_compensation=ac_comp,
                         normalize=normalize, relu=relu, overkill_factor=overkill)
            g['sub%d_out' % case] = np.asarray(mix._mag)
            g['sub%d_out_dtype' % case] = np.array(str(mix._mag.dtype))
            # invalidation (util_audio.py:149-157): ph kept, others cleared
            g['sub%d_state' % case] = np.array(
                [mix._wf is None, mix._F is None, mix._ph is not None,
                 mix._ref_mag is None, mix._D is None], dtype=np.int64)
            case += 1
    # raw-array subtrahend branch (util_audio.py:240-244)
    mix = make_ac(2048, 24, seed=300)
    raw = (np.random.default_rng(301).random((1025, 10)) ** 2).astype(np.float32)
    g['subraw_mix'] = mix._mag.copy()
    g['subraw_guess'] = raw.copy()
    mix.subtract(raw, offset=0.05)
    g['subraw_out'] = np.asarray(mix._mag)
    g['sub_cases'] = np.array(case)
    # offset beyond the window end: the reference raises (negative zeros dim,
    # util_audio.py:250-257) -- record that fact
    mix = make_ac(256, 40, seed=400)
    gs = make_ac(256, 13, seed=401)
    try:
        mix.subtract(gs, offset=10.0)
        g['sub_overrun_raises'] = np.array(0)
    except ValueError:
        g['sub_overrun_raises'] = np.array(1)

    # a10 _resize
    P = np.random.default_rng(7).random((6, 24)).astype(np.float32)
    for target in (8, 20):
        for t in (0, 1, 2, 3, 4, 5, 7, 8, 9, 19, 20, 21, 24):
            g['resize_%d_%d' % (t, target)] = np.asarray(
                ua.audio_complete._resize(P[:, :t], target))
    g['resize_P'] = P

    # a13 compress_bands
    for Fb in (1025, 2049):
        S = (np.random.default_rng(Fb).random((Fb, 6)) ** 2).astype(np.float32)
        g['cb_in_%d' % Fb] = S
        g['cb_out_%d' % Fb] = ua.audio_complete.compress_bands(S, bands=20)
    S = (np.random.default_rng(5).random((64, 3))).astype(np.float32)
    g['cb_lin_in'] = S
    g['cb_lin_out'] = ua.audio_complete.compress_bands(S, bands=8, log=False)

    # a9 section / slice / concat ; a11 resize ; a12 section_power
    ac = make_ac(512, 60, seed=11)
    g['sec_mag'] = ac._mag.copy()
    g['sec_ph'] = ac._ph.copy()
    sec = ac.section(0.2, None, 50)         # runs past the end -> zero padded
    g['sec_out_mag'] = np.asarray(sec._mag)
    g['sec_out_ph'] = np.asarray(sec._ph)
    g['sec_out_wf_len'] = np.array(len(sec._wf))
    sec2 = ac.section(0.1, 0.4)
    g['sec2_out_mag'] = np.asarray(sec2._mag)
    g['sec2_out_wf_len'] = np.array(len(sec2._wf))
    ac2 = ac.clone()
    ac2.slice(10, 40)
    g['slice_out_mag'] = np.asarray(ac2._mag)
    g['slice_out_wf_len'] = np.array(len(ac2._wf))
    ac2.concat(sec2)
    g['concat_out_mag'] = np.asarray(ac2._mag)
    g['concat_out_wf_len'] = np.array(len(ac2._wf))
    for i, (start, dur) in enumerate(((0.1, 0.2), (0.3, 0.02), (0.0, 0.5), (0.5, 0.01))):
        rs = ac.resize(start, dur, 8, attribs=['mag', 'ph'])
        g['rsz%d_args' % i] = np.array([start, dur])
        g['rsz%d_mag' % i] = np.asarray(rs._mag)
        g['rsz%d_ph' % i] = np.asarray(rs._ph)
        lo = ac.midi_tone_to_FFT(60)
        g['rsz%d_secpow' % i] = rs.section_power('mag', lo, lo + 348)
        g['rsz%d_secpow_hi' % i] = rs.section_power('mag', 200, 200 + 348)
    g['secpow_lo'] = np.array(lo)

    np.savez_compressed(OUT, **g)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes,', len(g), 'arrays')


if __name__ == '__main__':
    main()
