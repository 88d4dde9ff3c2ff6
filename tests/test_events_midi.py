"""CPU: note-event assembly (frames -> seconds as util_audio.py:269-272, overlap
de-duplication) and the Standard MIDI File writer (round trip through a reader)."""
import struct

import numpy as np

from amt_saga import events as E


def test_frames_to_seconds_matches_reference_vectors(refvec):
    for n_fft, T in ((2048, 516), (4096, 258), (4096, 130)):
        hop = n_fft // 4
        L = hop * (T - 1)
        fr = refvec['f2s_%d_%d_f' % (n_fft, T)]
        assert np.allclose(E.frames_to_seconds(fr, T, L, 44100), refvec['f2s_%d_%d' % (n_fft, T)],
                           rtol=1e-15, atol=0)


def test_events_to_notes_and_merge():
    T, L = 516, 512 * 515
    ev = np.array([[0, 0, 60, 0, 100, 100, 180],      # window 0: note at frame 100
                   [1, 0, 60, 0, 90, 100 - 258, 170], # (onset < 0 -> skipped)
                   [1, 0, 60, 0, 90, 0, 80],           # window 1 = +3 s
                   [1, 1, 64, 40, 80, 10, 5],          # end < onset -> one frame long
                   [2, 0, -1, -1, -1, -1, -1]], np.int32)
    notes = E.events_to_notes(ev, T, L)
    assert len(notes) == 3
    n0 = [n for n in notes if n['window'] == 0][0]
    assert abs(n0['start'] - 100 / 516 * L / 44100) < 1e-12
    short = [n for n in notes if n['pitch'] == 64][0]
    assert short['end'] > short['start'] and short['program'] == 40
    # the same piano C4 seen by two overlapping windows: 100 frames into window 0 vs
    # (3 s later window) frame 0 differ by > tol -> both kept
    assert len(E.merge_overlap_duplicates(notes)) == 3
    dup = E.events_to_notes(np.array([[0, 0, 60, 0, 100, 300, 350],
                                      [1, 0, 60, 0, 99, 42, 80]], np.int32), T, L)
    # window 1 starts at 2.9898 s; frame 300 of window 0 = 3.476 s, frame 42 of window 1 = 3.476 s
    merged = E.merge_overlap_duplicates(dup, merge_tol_s=0.02)
    assert len(merged) == 1 and merged[0]['end'] - merged[0]['start'] >= 38 / 516 * L / 44100 - 1e-9


def test_midi_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    notes = []
    for i in range(40):
        s = float(rng.uniform(0, 20))
        notes.append(dict(pitch=int(rng.integers(21, 109)), program=int(rng.choice([0, 24, 40, 111])),
                          velocity=int(rng.integers(1, 128)), start=s, end=s + float(rng.uniform(0.05, 2)),
                          window=0, iter=i))
    path = tmp_path / 'out.mid'
    E.write_midi(notes, str(path))
    raw = path.read_bytes()
    assert raw[:4] == b'MThd' and struct.unpack('>IHHH', raw[4:14]) == (6, 1, 5, 480)
    back = E.read_midi(str(path))
    assert len(back) == len(notes)
    key = lambda n: (round(n['start'] * 960), n['pitch'], n['program'])
    for a, b in zip(sorted(notes, key=key), sorted(back, key=key)):
        assert a['pitch'] == b['pitch'] and a['program'] == b['program'] and a['velocity'] == b['velocity']
        assert abs(a['start'] - b['start']) <= 0.5 / 960 + 1e-9
    # note-offs of overlapping same-pitch notes pair FIFO; the multiset of end times is preserved
    ends = lambda ns: sorted(round(n['end'] * 960) for n in ns)
    assert np.abs(np.array(ends(notes)) - np.array(ends(back))).max() <= 1
    assert E._vlq(0) == b'\x00' and E._vlq(0x80) == b'\x81\x00' and E._vlq(0x0FFFFFFF) == b'\xff\xff\xff\x7f'
