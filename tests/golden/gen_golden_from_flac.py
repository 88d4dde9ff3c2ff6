#!/usr/bin/env python3
"""Turn the reference's recorded subtraction fixtures into .npz golden vectors.

Run (this container only):  python tests/golden/gen_golden_from_flac.py
Reads  /root/reference/subtraction_demo/{name}_test{,_guess,_sub}.flac
Writes tests/golden/subtraction_demo_{name}.npz   (int32 PCM-24 data only)

The triples were produced by the reference's "short version with the
audio_util" cell (test_snippets.py:473-514): mixture and guess rendered by
fluidsynth, ``ac_sub.subtract(ac_guess, offset=0.5, ...)`` with N=4096, then
``.save()`` = librosa.istft -> soundfile PCM_24.  They are the only recorded
librosa outputs for the STFT -> magphase -> subtract -> iSTFT chain.

The per-scenario knobs are not recorded; a grid search over
normalize x attack_compensation in {-2..2} (kept below as ``search()``) finds
that {piano, strings-piano, overdriven} reproduce to <= 2.4e-7 absolute (two
PCM-24 LSBs) with normalize=True, attack_compensation=0, overkill_factor=1.
Those three are frozen as known-answer tests; the other scenarios were made
with hand-edited settings that the grid does not recover and are not used.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import flac_decode as fd   # noqa: E402

SRC = '/root/reference/subtraction_demo/'
FROZEN = {'piano': dict(normalize=True, attack_compensation=0),
          'strings-piano': dict(normalize=True, attack_compensation=0),
          'overdriven': dict(normalize=True, attack_compensation=0)}


def search(name):
    import itertools
    from oracle.audio import AudioCompleteOracle as AC
    mix, _ = fd.load_float(SRC + name + '_test.flac')
    g, _ = fd.load_float(SRC + name + '_test_guess.flac')
    sub, _ = fd.load_float(SRC + name + '_test_sub.flac')
    A = AC(mix, 4096); A.mag
    B = AC(g, 4096); B.mag
    best = None
    for norm, acomp in itertools.product((True, False), (-2, -1, 0, 1, 2)):
        a = A.clone()
        a.subtract(B.clone(), offset=0.5, attack_compensation=acomp, normalize=norm)
        m = np.abs(sub) < 0.999
        e = np.abs(a.wf - sub)[m].max()
        if best is None or e < best[0]:
            best = (e, norm, acomp)
    return best


def main():
    for name, knobs in FROZEN.items():
        out = {}
        for key, suffix in (('mix', '_test'), ('guess', '_test_guess'), ('sub', '_test_sub')):
            pcm, sr, bps = fd.decode(SRC + name + suffix + '.flac')
            assert sr == 44100 and bps == 24 and pcm.shape[1] == 1
            out[key] = pcm[:, 0].astype(np.int32)
        out['n_fft'] = np.array(4096)
        out['offset_s'] = np.array(0.5)
        out['normalize'] = np.array(int(knobs['normalize']))
        out['attack_compensation'] = np.array(knobs['attack_compensation'])
        path = os.path.join(HERE, 'subtraction_demo_%s.npz' % name)
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'search':
        for n in sys.argv[2:]:
            print(n, search(n))
    else:
        main()
