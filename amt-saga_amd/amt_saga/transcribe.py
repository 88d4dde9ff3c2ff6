"""Song-level driver over the batched loop: waveform (or FLAC) -> windows -> note events -> MIDI.

The reference only has this in its *training* form (training.py:296-449: 6-s windows advanced
by half a window, one note detected and subtracted per step, the gold note sequence standing
where predictions would).  This module composes the pieces the hot path and the widened rows
provide for the inference direction:

    flac.load_float / audio_from_file      util_audio.py:650-700 (file I/O)
    windows of p.timing_frames frames, hop = half a window        training.py:317-328
    TranscriptionLoop.run (all windows of the song in one batch)  the hot path
    events.events_to_notes -> merge_overlap_duplicates -> write_midi   util_audio.py:594-639, 790-792

    python -m amt_saga.transcribe in.flac out.mid [--weights DIR] [--iters 5]

Weights: a directory with {timing_start,timing_end,pitch,instrument,velocity}.npz in the
naming of amt_saga/rdcnn.py; without it the heads carry their seeded synthetic weights (the
reference ships no checkpoint), which exercises the whole path but transcribes nothing.
"""
import os
import sys

import numpy as np
import torch

from . import events as ev
from .hyperparams import Hyperparams
from .loop import TranscriptionLoop


def window_starts(n_samples, win_len, hop_len):
    """Start samples of the 50 %-overlapped windows covering the song (the last one is
    zero-padded)."""
    if n_samples <= win_len:
        return [0]
    n = 1 + int(np.ceil((n_samples - win_len) / hop_len))
    return [i * hop_len for i in range(n)]


def cut_windows(wf, win_len, hop_len):
    starts = window_starts(len(wf), win_len, hop_len)
    out = np.zeros((len(starts), win_len), dtype=np.float32)
    for i, s in enumerate(starts):
        seg = wf[s:s + win_len]
        out[i, :len(seg)] = seg
    return out, starts


def transcribe(wf, params=None, iters=5, heads=('timing', 'pitch', 'instrument', 'velocity'),
               groups=(0, 1, 2), weights_dir=None, guess='bank', loop=None, batch=1024):
    """wf: float32 mono waveform at params.sr.  Returns (notes, events) where notes is the
    merged list of dicts (pitch, program, velocity, start, end) and events the raw int32
    [iters, n_windows, 7] records."""
    p = params or Hyperparams(N=2048)
    if loop is None:
        loop = TranscriptionLoop(p, heads=heads, iters=iters, groups=groups, guess=guess)
        if weights_dir:
            for name, net in loop.nets.items():
                f = os.path.join(weights_dir, name + '.npz')
                if os.path.exists(f):
                    net.load_weights(f)
        loop.setup_device()
    L = p.H * (p.timing_frames - 1)
    wins, starts = cut_windows(np.asarray(wf, dtype=np.float32), L, L // 2)
    # song-level normalisers, as training.py:269-282 computes them: once per song, the maxima of the WHOLE song's
    # STFT and CQTs (one pass of the block-sum kernel over the song as a single signal), shared by every window
    song_refs = loop.song_levels(torch.from_numpy(np.ascontiguousarray(wf, dtype=np.float32)).cuda())
    # the loop proper: host batches streamed through run_stream (copy of batch i+1 under the compute of batch i)
    chunks = [wins[b0:b0 + batch] for b0 in range(0, len(wins), batch)]

    def refs_for(i):
        return {k: v.expand(len(chunks[i])).contiguous() for k, v in song_refs.items()}
    all_ev = [e.cpu().numpy() for e, _ in loop.run_stream(chunks, refs=refs_for)]
    evs = np.concatenate(all_ev, axis=1)
    notes = ev.events_to_notes(evs, p.timing_frames, L, sr=p.sr,
                               window_start_s=[s / p.sr for s in starts])
    return ev.merge_overlap_duplicates(notes), evs


def main(argv=None):
    import argparse
    from . import flac
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('infile')
    ap.add_argument('outfile')
    ap.add_argument('--weights', default=None)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--guess', default='bank', choices=('bank', 'render'))
    a = ap.parse_args(argv)
    wf, sr = flac.load_float(a.infile)
    if wf.ndim > 1:
        wf = wf.mean(axis=1)                     # [n, channels] -> mono
    notes, _ = transcribe(wf, Hyperparams(N=2048, sr=sr), iters=a.iters, weights_dir=a.weights, guess=a.guess)
    ev.write_midi(notes, a.outfile)
    print('%d notes -> %s' % (len(notes), a.outfile))


if __name__ == '__main__':
    main(sys.argv[1:])
