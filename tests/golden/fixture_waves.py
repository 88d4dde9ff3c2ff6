"""Test infrastructure: the INPUT side of the full-size parity fixtures (tests/golden/fullsize_fixtures.npz).

The fixtures hold what oracle/ computed in the build container (gen_fullsize_fixtures.py); the audio itself is not
stored (a 516-frame window is 1 MB) but re-rendered wherever a test runs, from seeded note lists through the float64
synthesiser restatement (oracle/synth.py), and then QUANTISED to 24-bit PCM (the reference's own file format,
util_audio.py:962-968): the float32 value q / 2^23 -- exact in float32 -- is what both sides read.  A last-ulp
difference between two machines' sin() moves a sample across a rounding boundary with probability ~1e-9, so the PCM is
the same everywhere; its SHA-1 is stored in the fixture and checked by the tests.  (16 bits would silence the quietest
windows: render()'s amplitude law (velocity / 128)^4 puts a velocity-5 note at 2e-6 of full scale.)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, 'amt-saga_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# name -> (hyperparameter keywords, heads, iterations, program groups, candidate windows, seed, notes per window, max onset)
CASES = {
    # BASELINE config C3 at the metric size: 516 frames, timing(start, end) + pitch + velocity, one subtraction
    'c3': dict(hp=dict(N=2048), heads=('timing', 'pitch', 'velocity'), iters=1, groups=(0,), seed=303, notes=(3, 3),
               max_onset=0.5, keep=16),
    # BASELINE config C5 as stated: all heads, five iterations, the three instrument groups, 516 frames
    'c5': dict(hp=dict(N=2048), heads=('timing', 'pitch', 'instrument', 'velocity'), iters=5, groups=(0, 1, 2), seed=505,
               notes=(2, 4), max_onset=0.5, keep=3),
    # the main small case of tests/test_gpu_loop.py (86 frames, 32 windows): only the oracle's OWN song-level
    # normalisers are stored; the oracle loop itself runs live in that test
    'main32': dict(hp=dict(N=2048, window_size_note_time=1), heads=('timing', 'pitch', 'velocity'), iters=2, groups=(0,),
                   seed=21, notes=(1, 3), max_onset=0.4, keep=32),
}


def params_for(case):
    from amt_saga.hyperparams import Hyperparams
    return Hyperparams(**CASES[case]['hp'])


def note_lists(case, n):
    """The first n seeded note lists of a case (host numbers only)."""
    from amt_saga import synth
    c = CASES[case]
    p = params_for(case)
    return synth.window_notes(n, c['seed'], c['notes'], c['groups'], c['max_onset'] * p.window_size_note_time)


def render_pcm(case, notes):
    """24-bit PCM as int32 [B, L] of the given note lists (L = hop x (frames - 1): a centred STFT gives exactly `frames`)."""
    from oracle import synth as osynth
    p = params_for(case)
    L = p.H * (p.timing_frames - 1)
    w = osynth.render_notes(notes, L, p.sr).astype(np.float64)
    return np.clip(np.rint(w * 8388608.0), -8388608, 8388607).astype(np.int32)


def pcm_to_wave(pcm):
    return (pcm.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)


def sha1(pcm):
    return hashlib.sha1(np.ascontiguousarray(pcm).tobytes()).hexdigest()
