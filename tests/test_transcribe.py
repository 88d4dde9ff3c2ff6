"""Song-level driver (amt_saga/transcribe.py): window cutting on the CPU; on the GPU a synthetic
clip goes FLAC -> windows -> batched loop -> note list -> MIDI and comes back well-formed."""
import os

import numpy as np
import pytest


def test_window_cutting():
    from amt_saga import transcribe as tr
    assert tr.window_starts(1000, 4000, 2000) == [0]
    assert tr.window_starts(4000, 4000, 2000) == [0]
    assert tr.window_starts(4001, 4000, 2000) == [0, 2000]
    assert tr.window_starts(10000, 4000, 2000) == [0, 2000, 4000, 6000]
    wf = np.arange(9000, dtype=np.float32)
    w, starts = tr.cut_windows(wf, 4000, 2000)
    assert starts == [0, 2000, 4000, 6000] and w.shape == (4, 4000)
    assert np.array_equal(w[1], wf[2000:6000])
    assert np.array_equal(w[3, :3000], wf[6000:]) and np.all(w[3, 3000:] == 0)


@pytest.mark.gpu
def test_transcribe_flac_to_midi(tmp_path):
    import torch
    assert torch.cuda.is_available()
    from amt_saga import synth, flac, events, transcribe as tr
    from amt_saga.hyperparams import Hyperparams
    p = Hyperparams(N=2048, window_size_note_time=1)           # 1-s windows keep the test short
    L = p.H * (p.timing_frames - 1)
    n = int(3.2 * L)
    notes_in = [(0, 60, 100, 0.2, 0.5), (0, 64, 90, 0.9, 0.4), (1, 67, 80, 1.6, 0.6), (2, 72, 110, 2.4, 0.3)]
    from oracle import synth as osynth
    wf = osynth.render_window(notes_in, n, p.sr).numpy()
    path = str(tmp_path / 'clip.flac')
    flac.save_float(wf, path, p.sr)
    back, sr = flac.load_float(path)
    assert sr == p.sr and np.abs(back - wf).max() < 2.0 ** -22
    notes, evs = tr.transcribe(back, p, iters=2, heads=('timing', 'pitch', 'instrument', 'velocity'))
    n_win = len(tr.window_starts(n, L, L // 2))
    assert evs.shape == (2, n_win, 7)
    assert np.array_equal(evs[0, :, 0], np.arange(n_win))       # window ids follow the song order
    assert all(21 <= e['pitch'] <= 108 and e['end'] > e['start'] >= 0 for e in notes)
    assert len(notes) <= 2 * n_win
    # the same song in batches of 3 windows (song-level normalisers over all batches, batches streamed with the
    # copy overlapped): the events do not depend on how the windows are batched
    notes_b, evs_b = tr.transcribe(back, p, iters=2, heads=('timing', 'pitch', 'instrument', 'velocity'), batch=3)
    assert np.array_equal(evs_b, evs) and notes_b == notes
    # song-level constants: the product's whole-song maxima vs the oracle's (training.py:269-282)
    from amt_saga.loop import TranscriptionLoop
    from oracle import audio as oa, cqt as ocqt
    lv = TranscriptionLoop(p, heads=('timing', 'pitch'), iters=1).song_levels(torch.from_numpy(back).cuda())
    ref_mag = oa.magphase(oa.stft(back, p.N, p.H))[0].max()
    tab = ocqt.cqt_table(p.sr, float(oa.midi_to_hz(p.pitch_low)), p.pitch_high - p.pitch_low, 12)
    ref_c1 = ocqt.cqt_window_max(back, tab[0], tab[1], p.H)
    assert abs(float(lv['ref_mag']) - ref_mag) <= 1e-4 * ref_mag
    assert abs(float(lv['ref_C_1']) - ref_c1) <= 1e-4 * ref_c1
    mid = str(tmp_path / 'out.mid')
    events.write_midi(notes, mid)
    rd = events.read_midi(mid)
    assert len(rd) == len(notes)
    assert sorted(e['pitch'] for e in rd) == sorted(e['pitch'] for e in notes)
    # the command-line entry does the same
    tr.main([path, str(tmp_path / 'cli.mid'), '--iters', '1'])
    assert os.path.getsize(str(tmp_path / 'cli.mid')) > 20
