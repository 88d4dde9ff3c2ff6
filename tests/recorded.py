"""Loader for the reference's recorded fixtures (tests/golden/recorded_waves.npz +
recorded_index.json, produced by tests/golden/gen_golden_from_flac.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SCALE = 1.0 / (1 << 23)                 # libsndfile float scaling of PCM-24
FULL = (1 << 23) - 1

_cache = {}


def index():
    if 'index' not in _cache:
        with open(os.path.join(GOLDEN, 'recorded_index.json')) as f:
            _cache['index'] = json.load(f)
    return _cache['index']


def wave(key):
    """int32 PCM-24 samples of one recorded file."""
    if 'waves' not in _cache:
        _cache['waves'] = np.load(os.path.join(GOLDEN, 'recorded_waves.npz'))
    return _cache['waves']['w_' + key]


def frozen_triples():
    return sorted(k for k, v in index()['triples'].items() if v['frozen'])


def triple(name):
    """dict(mix, guess, sub: int32 PCM; normalize, attack_compensation, overkill_factor,
    guess_ref_mag_correction, offset_frames, offset_s, n_fft)."""
    r = dict(index()['triples'][name])
    r.update(mix=wave(r['mix']), guess=wave(r['guess']), sub=wave(r['sub']),
             n_fft=index()['n_fft'], hop=index()['hop'], offset_s=index()['offset_s'])
    return r


def reproducible_mask(mix_i, guess_i, sub_i, offset_frames, n_fft=4096, hop=1024):
    """Output samples whose recorded value follows from the recorded inputs: not clipped by the
    PCM-24 writer and out of reach (n_fft + hop) of any input sample the writer clipped."""
    m = np.abs(sub_i) < FULL - 1
    reach = n_fft + hop
    for j in np.flatnonzero((guess_i >= FULL) | (guess_i <= -FULL - 1)):
        c = offset_frames * hop + int(j)
        m[max(c - reach, 0):c + reach] = False
    for j in np.flatnonzero((mix_i >= FULL) | (mix_i <= -FULL - 1)):
        m[max(int(j) - reach, 0):int(j) + reach] = False
    return m


def run_triple(AC, name):
    """The recorded scenario through an audio_complete implementation `AC` (oracle or product):
    returns (resynthesised residual waveform, recorded residual as float, mask, scenario)."""
    z = triple(name)
    a = AC(z['mix'] * SCALE, z['n_fft'])
    g = AC(z['guess'] * SCALE, z['n_fft'])
    if z['guess_ref_mag_correction'] != 1.0:
        # the guess FILE is clipped inside its loudest frame; the reference normalised with the
        # ref_mag of the unclipped waveform (fitted once by gen_golden_from_flac.py)
        g._ref_mag = np.float32(g.ref_mag * z['guess_ref_mag_correction'])
    a.subtract(g, offset=z['offset_s'], attack_compensation=z['attack_compensation'],
               normalize=z['normalize'], overkill_factor=z['overkill_factor'])
    m = reproducible_mask(z['mix'], z['guess'], z['sub'], z['offset_frames'], z['n_fft'], z['hop'])
    return a.wf, z['sub'] * SCALE, m, z
