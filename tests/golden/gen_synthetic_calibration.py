#!/usr/bin/env python3
"""Calibrate the synthetic head weights so that they behave like trained ones.

Run (CPU, this container or any host):  python tests/golden/gen_synthetic_calibration.py
Writes amt-saga_amd/amt_saga/data/synthetic_heads.npz

Why.  The reference ships no weights, so the heads carry seeded random ones.  With random
BatchNorm statistics a 33-layer sigmoid stack forgets its input: every window gets the same pitch / onset,
and "bit-exact predicted indices" is then tested on a constant.  A trained Keras model's BN layers hold the
running mean / variance of their own inputs; this script gives the synthetic heads that property
(data-dependent initialisation): the convolution / dense kernels stay what ``res_net.init_weights(seed)``
draws, and layer by layer, on a calibration batch of the features the loop really feeds the head,
  * every BatchNormalization gets moving_mean / moving_variance = the batch statistics of its input
    (conv output, projected shortcut, residual sum), gamma ~ U(0.8, 2.0) / U(0.8, 1.2), beta ~ N(0, 0.3);
  * the last Dense is rescaled so that the head's outputs spread over its range across windows
    (regression heads: logit mean / std per head below; softmax heads: per-class centred, std 2.5).
Only those small tensors (BN parameters, last Dense) are stored, keyed by topology signature + seed;
``init_weights`` overlays them on the seeded draw.  The forward used here is the numpy oracle's own
operators (oracle/rdcnn.py) -- this is offline fixture generation, like training would be, and nothing
in the product path executes it.

Calibration features come from the CPU synthesiser (oracle.synth.render_window) through the oracle's
feature recipe: C_timing for the timing heads; CQT slices at random (onset, end) for pitch / instrument /
velocity, of the window and of a partly subtracted residual.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]

from oracle import audio as oa, cqt as ocqt, rdcnn as orc, synth as osynth   # noqa: E402
from amt_saga import heads as H, synth                              # noqa: E402
from amt_saga.hyperparams import Hyperparams                        # noqa: E402
from amt_saga.rdcnn import topology_signature                       # noqa: E402

OUT = os.path.join(ROOT, 'amt-saga_amd', 'amt_saga', 'data', 'synthetic_heads.npz')
N_CAL = 24                                   # calibration windows per head
# regression heads: (mean, std) of the last Dense's output over the calibration batch
LOGIT_TARGET = {'timing_start': (-0.9, 0.7), 'timing_end': (0.5, 0.7), 'timing': (0.0, 1.0),
                'pitch': (0.0, 1.2), 'velocity': (0.0, 1.2)}


def bn_stats(x):
    return x.mean(axis=(0, 1, 2)).astype(np.float32), x.var(axis=(0, 1, 2)).astype(np.float32)


def calibrate(w, cfg, xs, role, seed):
    """Walk the graph of oracle.rdcnn.forward, filling BN statistics from the batch as it goes."""
    rng = np.random.default_rng(seed + 77000)
    w = {k: np.array(v, dtype=np.float32) for k, v in w.items()}
    out = {}

    def fit_bn(x, prefix, glo, ghi):
        c = x.shape[-1]
        m, v = bn_stats(x)
        for k, a in (('gamma', rng.uniform(glo, ghi, c)), ('beta', rng.normal(0, 0.3, c)), ('mean', m), ('var', v)):
            w[prefix + '/' + k] = out[prefix + '/' + k] = np.asarray(a, np.float32)
        return orc.batchnorm(x, w, prefix)

    r = cfg['residual_layer_frequencies'][0] if cfg['residual_layer_frequencies'] else 0
    flats = []
    for t, x in enumerate(xs):
        p1 = np.asarray(x, np.float32)
        p0 = p1
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            p1 = orc.conv2d_same(p1, w['t%d/conv%d/kernel' % (t, i)], w['t%d/conv%d/bias' % (t, i)])
            p1 = orc.sigmoid(fit_bn(p1, 't%d/bn%d' % (t, i), 0.8, 2.0))
            if r and i % r == 0:
                a = p0
                if a.shape != p1.shape:
                    if a.shape[-1] != p1.shape[-1]:
                        a = a @ w['t%d/sc%d/kernel' % (t, i)][0, 0] + w['t%d/sc%d/bias' % (t, i)]
                    if a.shape[1:3] != p1.shape[1:3]:
                        a = orc.pool2d(a, (a.shape[1] // p1.shape[1], a.shape[2] // p1.shape[2]), 'avg')
                    a = fit_bn(a, 't%d/scbn%d' % (t, i), 0.8, 1.2)
                p1 = fit_bn(a + p1, 't%d/resbn%d' % (t, i), 0.8, 1.2)
                p0 = p1
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1 = orc.pool2d(p1, cfg['pool_sizes'][t], 'max')
        flats.append(p1.reshape(p1.shape[0], -1))
    flat = np.concatenate(flats, axis=1)
    h = orc.sigmoid(flat @ w['dense1/kernel'] + w['dense1/bias'])
    z = h @ w['dense2/kernel']
    if cfg['output_classes'] == 1:
        mean, std = LOGIT_TARGET[role]
        sc = std / max(float(z.std()), 1e-12)
        out['dense2/kernel'] = (w['dense2/kernel'] * sc).astype(np.float32)
        out['dense2/bias'] = np.array([mean - float(z.mean()) * sc], np.float32)
    else:
        zc = z - z.mean(axis=0, keepdims=True)
        sc = 2.5 / max(float(zc.std()), 1e-12)
        out['dense2/kernel'] = (w['dense2/kernel'] * sc).astype(np.float32)
        out['dense2/bias'] = (-z.mean(axis=0) * sc).astype(np.float32)
    w.update(out)
    y = orc.forward(w, cfg, xs)
    return out, y


def windows(p, n, seed, groups):
    L = p.H * (p.timing_frames - 1)
    notes = synth.window_notes(n, seed, (1, 4), groups, 0.5 * p.window_size_note_time)
    return osynth.render_notes(notes, L, p.sr)


def timing_features(p, n, seed):
    rng = np.random.default_rng(seed)
    feats = []
    for wv in windows(p, n, seed, (0, 1, 2)):
        mag = oa.magphase(oa.stft(wv, p.N, p.H))[0]
        ref = mag.max()
        # half of the batch looks like a later iteration: part of the window already subtracted
        if rng.random() < 0.5:
            t0 = int(rng.integers(0, mag.shape[1] - 2))
            mag = mag.copy()
            mag[:, t0:t0 + int(rng.integers(20, 200))] *= rng.uniform(0.05, 0.6)
        ct = oa.AudioCompleteOracle._resize(oa.AudioCompleteOracle.compress_bands(mag, p.timing_bands),
                                            p.timing_frames) / ref
        feats.append(ct.astype(np.float32))
    return [np.stack(feats)[..., None]]


def cqt_features(p, n, seed, kind):
    rng = np.random.default_rng(seed)
    f_lo = float(oa.midi_to_hz(p.pitch_low))
    span = p.pitch_high - p.pitch_low
    feats = []
    for wv in windows(p, n, seed, (0, 1, 2)):
        T = p.timing_frames
        s = int(rng.integers(0, T - 12))
        e = s + int(rng.integers(2, 120))
        src = ocqt.slice_C_frames(T, s, e, p.pitch_frames)
        if kind == 'pitch':
            tab = ocqt.cqt_table(p.sr, f_lo, p.pitch_bands, 12 * p.pitch_bins_per_tone)
            rt = ocqt.cqt_table(p.sr, f_lo, span, 12)
        elif kind == 'instrument':
            tab = ocqt.cqt_table(p.sr, f_lo, p.instrument_bands, 12 * p.instrument_bins_per_tone)
            rt = ocqt.cqt_table(p.sr, f_lo, span * p.instrument_bins_per_tone, 12 * p.instrument_bins_per_tone)
        else:
            pitch = int(rng.integers(p.pitch_low, p.pitch_high + 1))
            tab = ocqt.cqt_table(p.sr, float(oa.midi_to_hz(pitch - 10)), p.bins_velocity, 24)
            rt = ocqt.cqt_table(p.sr, f_lo, span * p.instrument_bins_per_tone * 4,
                                12 * p.instrument_bins_per_tone * 4)
        # the normaliser (max of the whole CQT) on a coarse sub-grid of its bins: calibration needs the
        # scale, not the exact maximum
        sub = slice(None, None, max(1, len(rt[0]) // 87))
        ref = ocqt.cqt_window_max(wv, rt[0][sub], rt[1][sub], p.H)
        c = ocqt.cqt_frames(wv, src, tab[0], tab[1], p.H) / max(ref, 1e-12)
        if rng.random() < 0.4:
            c = c * rng.uniform(0.1, 0.7)                      # a residual after subtraction is quieter
        feats.append(c.astype(np.float32))
    return np.stack(feats)[..., None]


def main():
    jobs = []
    p6 = Hyperparams(N=2048)
    p1 = Hyperparams(N=2048, window_size_note_time=1)
    p4 = Hyperparams(N=4096)
    for p, tag in ((p6, 'N2048/6s'), (p1, 'N2048/1s'), (p4, 'N4096/6s')):
        for role, seed in (('timing_start', 107), ('timing_end', 105), ('timing', 104)):
            jobs.append((tag, role, 'timing', p, seed))
    for role, seed in (('pitch', 101), ('instrument', 102), ('instrument_dual', 102), ('velocity', 103)):
        jobs.append(('any', role, role, p1, seed))
    store = {}
    cache = {}
    for tag, role, kind, p, seed in jobs:
        if kind == 'timing':
            head = H.timming_classifier(p, weight_seed=seed, calibrated=False)
            key = ('timing', tag)
            if key not in cache:
                cache[key] = timing_features(p, N_CAL, 9000 + p.timing_frames)
            xs = cache[key]
        elif kind == 'pitch':
            head = H.pitch_classifier(p, weight_seed=seed, calibrated=False)
            xs = [cqt_features(p, N_CAL, 9101, 'pitch')]
        elif kind == 'velocity':
            head = H.VelocityClassifier(p, weight_seed=seed, calibrated=False)
            xs = [cqt_features(p, N_CAL, 9103, 'velocity')]
        else:
            head = H.InstrumentClassifier(p, kind, weight_seed=seed, calibrated=False)
            if ('inst',) not in cache:
                cache[('inst',)] = cqt_features(p, N_CAL, 9102, 'instrument')
            xs = [cache[('inst',)]] * len(head.cfg['input_shapes'])
            if len(xs) == 2:                                   # second tower: the linear-FFT style input, here a shuffled batch
                xs = [xs[0], xs[0][::-1].copy()]
        sig = topology_signature(head.cfg, seed)
        out, y = calibrate(head.weights, head.cfg, xs, role, seed)
        for k, v in out.items():
            store[sig + '/' + k] = v
        if head.cfg['output_classes'] == 1:
            print('%-10s %-16s seed %d  %s  outputs min %.1f med %.1f max %.1f' %
                  (tag, role, seed, sig, y.min(), np.median(y), y.max()), flush=True)
        else:
            print('%-10s %-16s seed %d  %s  distinct argmax %d of %d windows' %
                  (tag, role, seed, sig, len(set(np.argmax(y, 1).tolist())), len(y)), flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **store)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes,', len(store), 'tensors')


if __name__ == '__main__':
    main()
