#!/usr/bin/env python3
"""Test infrastructure: full-size parity fixtures, computed by oracle/ in the BUILD CONTAINER (no GPU) and committed.

    python tests/golden/gen_fullsize_fixtures.py [case ...]        (default: every case; ~10 minutes on 8 cores)

The oracle loop needs seconds per BASELINE-sized window and its song-level normalisers (`LoopOracle.ref_levels`: every
bin of the 87- / 348- / 1392-bin grids over all 516 frames, training.py:271-282) half a minute, so the GPU tests cannot
run them live at the metric size.  This script runs them here, once, and stores what the product must reproduce
(tests/test_gpu_fullsize_fixtures.py):

  * c3     config C3 at 516 frames, 16 windows: timing(start, end) + pitch + velocity, one subtraction
           (training.py:296-449 with the predicted note);
  * c5     config C5 as stated: all heads, five iterations, instrument groups 0-2, 516 frames, 3 windows;
  * main32 the oracle's own normalisers for the 32 small windows of tests/test_gpu_loop.py's main case.

Per kept window: the oracle's OWN normalisers (ref_mag, ref_C_1, ref_C_inst, ref_C_foc -- the product's prepare() is
compared with them, and the oracle loop ran on them, not on the product's), the integer events, the heads'
pre-rounding floats, and the residual: 16-bit quantised in full for the first windows (step 1.5e-5 of the maximum; the
bar is 1e-4) and as per-frame maxima + the 20-band compression (util_audio.py:436-466) for all of them.
A fixture cannot hand a near-tie decision over the way the live comparison does (oracle/compare.py), so candidate
windows with any decision closer than 10 x the head's band to a rounding boundary are not kept (the count is stored).
The audio is re-rendered from seeded note lists and 24-bit quantised on both sides (fixture_waves.py)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixture_waves as fw                                   # noqa: E402  (also puts the repo on sys.path)

from oracle import audio as oa                               # noqa: E402
from oracle import synth as osynth                           # noqa: E402
from oracle.compare import FLOAT_TOL                         # noqa: E402
from oracle.loop import LoopOracle                           # noqa: E402

REF_KEYS = ('ref_mag', 'ref_C_1', 'ref_C_inst', 'ref_C_foc')
HEAD_KEYS = ('timing_start', 'timing_end', 'pitch', 'instrument', 'velocity')
MARGIN = 10.0                                                # x band
FULL_RESID = {'c3': 2, 'c5': 3}
_ORC = None


def build_oracle(case):
    from amt_saga import synth
    from amt_saga.loop import TranscriptionLoop
    c = fw.CASES[case]
    p = fw.params_for(case)
    lp = TranscriptionLoop(p, heads=c['heads'], iters=c['iters'], groups=c['groups'])      # host side only: the seeded weights
    bank = osynth.guess_bank_waves(c['groups'], p.pitch_low, p.pitch_high, sr=p.sr)
    remap = np.zeros(3, np.int32)
    for i, g in enumerate(c['groups']):
        remap[g] = i
    return LoopOracle(p, c['heads'], {k: n.weights for k, n in lp.nets.items()}, iters=c['iters'],
                      prog_group=remap[synth.prog_group_table(p.instrument_classes)], bank_waves=bank)


def _refs_only(wave):
    r = _ORC.ref_levels(wave)
    return [float(r.get(k, np.nan)) for k in REF_KEYS]


def _window(wave):
    orc = _ORC
    refs = orc.ref_levels(wave)
    ev, mag = orc.run_window(wave, {k: float(v) for k, v in refs.items()}, 0)
    worst = np.inf                                           # smallest margin / band over the window's decisions
    floats = {}
    for name, it, y, margin, v, forced in orc.decisions:
        worst = min(worst, margin / FLOAT_TOL[name])
        floats.setdefault(name, []).append(np.atleast_1d(np.asarray(y, np.float32)))
    floats = {k: np.stack(v) for k, v in floats.items()}     # [iters, K]
    band = oa.AudioCompleteOracle.compress_bands(mag, bands=orc.p.timing_bands).astype(np.float32)
    return dict(refs=[float(refs.get(k, np.nan)) for k in REF_KEYS], events=ev, floats=floats, mag=mag.astype(np.float32),
                band=band, fmax=mag.max(axis=0).astype(np.float32), worst=worst)


def run_case(case, out, workers):
    global _ORC
    c = fw.CASES[case]
    keep = c['keep']
    t0 = time.time()
    _ORC = build_oracle(case)
    print(case, 'oracle built in %.0f s' % (time.time() - t0), flush=True)
    if case == 'main32':
        pcm = fw.render_pcm(case, fw.note_lists(case, keep))
        with mp.get_context('fork').Pool(workers) as pool:
            refs = pool.map(_refs_only, list(fw.pcm_to_wave(pcm)))
        out[case + '_refs'] = np.asarray(refs, np.float32)
        out[case + '_sha1'] = np.asarray(fw.sha1(pcm))
        print(case, 'done in %.0f s' % (time.time() - t0), flush=True)
        return
    n_cand = keep + max(4, keep // 2)
    notes = fw.note_lists(case, n_cand)
    pcm = fw.render_pcm(case, notes)
    with mp.get_context('fork').Pool(workers) as pool:
        res = pool.map(_window, list(fw.pcm_to_wave(pcm)), chunksize=1)
    idx = [i for i, r in enumerate(res) if r['worst'] >= MARGIN][:keep]
    if len(idx) < keep:
        raise SystemExit('%s: only %d of %d candidates clear the tie margin' % (case, len(idx), n_cand))
    rejected = [i for i in range(idx[-1] + 1) if i not in idx]
    sel = [res[i] for i in idx]
    out[case + '_idx'] = np.asarray(idx, np.int32)
    out[case + '_rejected_near_tie'] = np.asarray(rejected, np.int32)
    out[case + '_sha1'] = np.asarray(fw.sha1(pcm[idx]))
    out[case + '_refs'] = np.asarray([r['refs'] for r in sel], np.float32)
    ev = np.stack([r['events'] for r in sel], axis=1)                       # [iters, keep, 7]
    ev[:, :, 0] = np.arange(keep)[None, :]
    out[case + '_events'] = ev.astype(np.int32)
    for h in HEAD_KEYS:
        if h in sel[0]['floats']:
            out[case + '_float_' + h] = np.stack([r['floats'][h] for r in sel], axis=1)    # [iters, keep, K]
    out[case + '_band'] = np.stack([r['band'] for r in sel])
    out[case + '_fmax'] = np.stack([r['fmax'] for r in sel])
    nfull = FULL_RESID[case]
    scale = np.asarray([r['mag'].max() / 65535.0 for r in sel[:nfull]], np.float64)
    out[case + '_resid_scale'] = scale
    out[case + '_resid_q'] = np.stack([np.rint(r['mag'] / s).astype(np.uint16) for r, s in zip(sel[:nfull], scale)])
    out[case + '_worst_margin_over_band'] = np.asarray([r['worst'] for r in sel], np.float32)
    print(case, 'kept', idx, 'rejected near a tie', rejected, 'done in %.0f s' % (time.time() - t0), flush=True)


if __name__ == '__main__':
    cases = sys.argv[1:] or list(fw.CASES)
    path = os.path.join(HERE, 'fullsize_fixtures.npz')
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for case in cases:
        for k in [k for k in out if k.startswith(case + '_')]:
            del out[k]
        run_case(case, out, workers=min(8, os.cpu_count() or 1))
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f MB' % (os.path.getsize(path) / 1e6))
