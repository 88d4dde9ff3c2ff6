"""Note-event assembly and Standard MIDI File output.

The loop emits fixed-size records per window and iteration
    {window, iter, pitch, program, velocity, onset_frame, end_frame}      (int32 x 7)
This module turns them into a song-level note list and writes it as a MIDI
file -- the step the reference does with ``note_sequence.add_note`` / ``save``
(util_audio.py:594-639, 790-792; magenta / pretty_midi, neither available here)
after the sliding-window advance of training.py:317-328 (6-s windows moved by
half a window).  Host-side, plain Python: a few hundred integers per song.

  * frames -> seconds is the reference's own map (util_audio.py:269-272):
        t = frames / T / sr * len(wf)
    shifted by the window's start inside the song;
  * notes that the 50 %-overlapped windows both report (same pitch and program,
    onsets within `merge_tol_s`) are merged, keeping the longer one;
  * the file is SMF format 1, 480 ticks per quarter at 120 bpm (1 tick = 1/960 s),
    one track per program (channels 0-15 round-robin, skipping 9 = drums).
"""
import struct

import numpy as np

TICKS_PER_QUARTER = 480
TEMPO_US_PER_QUARTER = 500000            # 120 bpm
TICKS_PER_SECOND = TICKS_PER_QUARTER * 1e6 / TEMPO_US_PER_QUARTER


def frames_to_seconds(frames, n_frames, n_samples, sr):
    """audio_complete._frames_to_seconds (util_audio.py:269-272)."""
    return np.asarray(frames, dtype=np.float64) / n_frames / sr * n_samples


def events_to_notes(events, n_frames, n_samples, sr=44100, window_start_s=None, hop_windows_s=None):
    """events: int array [..., 7] (any leading shape).  window_start_s[w] gives the
    start of window w in the song (default: w * hop_windows_s, default hop = half a
    window, training.py:317-328).  Returns a list of dicts sorted by start time."""
    ev = np.asarray(events).reshape(-1, 7)
    win_len_s = n_samples / sr
    if hop_windows_s is None:
        hop_windows_s = win_len_s / 2.0
    notes = []
    for w, it, pitch, program, velocity, on, off in ev:
        if pitch < 0 or on < 0:
            continue
        t0 = float(window_start_s[w]) if window_start_s is not None else w * hop_windows_s
        start = t0 + float(frames_to_seconds(on, n_frames, n_samples, sr))
        end = t0 + float(frames_to_seconds(max(off, on + 1), n_frames, n_samples, sr))
        notes.append(dict(pitch=int(pitch), program=int(max(program, 0)),
                          velocity=int(velocity) if velocity > 0 else 64,
                          start=start, end=end, window=int(w), iter=int(it)))
    notes.sort(key=lambda n: (n['start'], n['pitch'], n['program']))
    return notes


def merge_overlap_duplicates(notes, merge_tol_s=0.05):
    """Drop the second report of a note seen by two overlapping windows."""
    out = []
    for n in sorted(notes, key=lambda n: (n['program'], n['pitch'], n['start'])):
        if out:
            m = out[-1]
            if (m['program'] == n['program'] and m['pitch'] == n['pitch'] and
                    m['window'] != n['window'] and abs(m['start'] - n['start']) <= merge_tol_s):
                if n['end'] - n['start'] > m['end'] - m['start']:
                    out[-1] = n
                continue
        out.append(n)
    out.sort(key=lambda n: (n['start'], n['pitch'], n['program']))
    return out


def _vlq(v):
    v = int(v)
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def write_midi(notes, path):
    """Write the note list as a format-1 Standard MIDI File."""
    programs = sorted({n['program'] for n in notes})
    chans = [c for c in range(16) if c != 9]
    tracks = [b'\x00\xff\x51\x03' + struct.pack('>I', TEMPO_US_PER_QUARTER)[1:] + b'\x00\xff\x2f\x00']
    for i, prog in enumerate(programs):
        ch = chans[i % len(chans)]
        evs = []
        for n in notes:
            if n['program'] != prog:
                continue
            t_on = int(round(n['start'] * TICKS_PER_SECOND))
            t_off = max(int(round(n['end'] * TICKS_PER_SECOND)), t_on + 1)
            evs.append((t_on, 1, bytes([0x90 | ch, n['pitch'] & 0x7F, min(max(n['velocity'], 1), 127)])))
            evs.append((t_off, 0, bytes([0x80 | ch, n['pitch'] & 0x7F, 0])))
        evs.sort(key=lambda e: (e[0], e[1]))
        data = b'\x00' + bytes([0xC0 | ch, prog & 0x7F])
        last = 0
        for t, _, msg in evs:
            data += _vlq(t - last) + msg
            last = t
        data += b'\x00\xff\x2f\x00'
        tracks.append(data)
    with open(path, 'wb') as f:
        f.write(b'MThd' + struct.pack('>IHHH', 6, 1, len(tracks), TICKS_PER_QUARTER))
        for t in tracks:
            f.write(b'MTrk' + struct.pack('>I', len(t)) + t)


def _read_vlq(buf, i):
    v = 0
    while True:
        b = buf[i]; i += 1
        v = (v << 7) | (b & 0x7F)
        if not b & 0x80:
            return v, i


def read_midi(path):
    """Standard MIDI File reader for the note content note_sequence.save() / write_midi produce:
    format 0/1, ticks-per-quarter division, set-tempo meta events (the first tempo applies to the whole
    file, as in the single-tempo files magenta writes), running status, note-on with velocity 0 as
    note-off, program changes.  Returns notes sorted by start time (seconds)."""
    data = open(path, 'rb').read()
    if data[:4] != b'MThd':
        raise ValueError('not a Standard MIDI File')
    _, fmt, ntr, div = struct.unpack('>IHHH', data[4:14])
    if div & 0x8000:
        raise ValueError('SMPTE time division is not supported')
    pos = 14
    raw_notes = []
    tempo = None
    for _ in range(ntr):
        if data[pos:pos + 4] != b'MTrk':
            raise ValueError('missing MTrk chunk')
        ln = struct.unpack('>I', data[pos + 4:pos + 8])[0]
        trk = data[pos + 8:pos + 8 + ln]
        pos += 8 + ln
        i, t, status = 0, 0, 0
        prog = {}
        open_notes = {}
        while i < len(trk):
            d, i = _read_vlq(trk, i)
            t += d
            if trk[i] & 0x80:
                status = trk[i]; i += 1
            if status == 0xFF:                                   # meta event
                typ = trk[i]
                ln2, i = _read_vlq(trk, i + 1)
                if typ == 0x51 and tempo is None:
                    tempo = int.from_bytes(trk[i:i + 3], 'big')
                i += ln2
                continue
            if status in (0xF0, 0xF7):                           # sysex
                ln2, i = _read_vlq(trk, i)
                i += ln2
                continue
            hi, ch = status & 0xF0, status & 0x0F
            if hi == 0xC0:
                prog[ch] = trk[i]; i += 1
            elif hi == 0xD0:
                i += 1
            elif hi in (0x90, 0x80):
                p, v = trk[i], trk[i + 1]; i += 2
                if hi == 0x90 and v > 0:
                    open_notes.setdefault((ch, p), []).append((t, v))
                elif open_notes.get((ch, p)):
                    t0, v0 = open_notes[(ch, p)].pop(0)     # overlapping same-pitch notes: FIFO
                    raw_notes.append((p, prog.get(ch, 0), v0, t0, t))
            else:
                i += 2
    us_per_tick = (tempo if tempo is not None else TEMPO_US_PER_QUARTER) / float(div)
    notes = [dict(pitch=p, program=pr, velocity=v, start=t0 * us_per_tick * 1e-6, end=t1 * us_per_tick * 1e-6)
             for p, pr, v, t0, t1 in raw_notes]
    notes.sort(key=lambda n: (n['start'], n['pitch'], n['program']))
    return notes
