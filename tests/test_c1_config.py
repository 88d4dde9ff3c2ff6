"""BASELINE config C1: a single window of the reference's own piano_test.flac
(tests/golden/recorded_waves.npz), 2048-pt STFT + pitch_classifier.
CPU: the oracle path (plumbing).  GPU: the drop-in single-window API
(audio_complete + pitch_classifier.classify) against the oracle."""
import os

import numpy as np
import pytest

from oracle import audio as oa, cqt as ocqt, params as op, rdcnn as orc


def _window(golden_dir):
    import recorded as rec
    return (rec.triple('piano')['mix'] / float(1 << 23)).astype(np.float32)


def _oracle_c1(wf, weights, p, onset=0.5, dur=1.0):
    ac = oa.AudioCompleteOracle(wf, p.N, p.H)
    mag = ac.mag
    s, t = ac._seconds_to_frames(onset), ac._seconds_to_frames(onset + dur)
    src = ocqt.slice_C_frames(mag.shape[1], s, t, p.pitch_frames)
    tab = ocqt.cqt_table(p.sr, float(oa.midi_to_hz(p.pitch_low)), p.pitch_bands, 12 * p.pitch_bins_per_tone)
    C = ocqt.cqt_frames(wf, src, tab[0], tab[1], p.H)
    ref = C.max()
    y = orc.forward(weights, orc.head_config(p, 'pitch'), [(C / ref).astype(np.float32)[None, :, :, None]])
    return mag, C, y


def test_c1_oracle_plumbing(golden_dir):
    from amt_saga.heads import pitch_classifier
    from amt_saga.hyperparams import Hyperparams
    p = Hyperparams(N=2048)
    wf = _window(golden_dir)
    assert wf.shape == (132300,)
    head = pitch_classifier(p)                          # host-side object: weights only, no GPU touched
    mag, C, y = _oracle_c1(wf, head.weights, op.HyperparamsOracle(N=2048))
    assert mag.shape == (1025, 259) and C.shape == (174, 8) and y.shape == (1, 1)
    assert 21 <= float(y[0, 0]) <= 108
    # the chord G3-B3-D4 + C2 is in the window: the CQT slice peaks on one of those pitches
    k = int(np.argmax(C.max(axis=1)))
    midi = 21 + k / 2.0
    assert min(abs(midi - m) for m in (36, 55, 59, 62, 48, 67, 74)) <= 0.5


@pytest.mark.gpu
def test_c1_hip_vs_oracle(golden_dir):
    from amt_saga.audio import audio_complete
    from amt_saga.heads import pitch_classifier
    from amt_saga.hyperparams import Hyperparams
    p = Hyperparams(N=2048)
    wf = _window(golden_dir)
    head = pitch_classifier(p)
    mag_ref, C_ref, y_ref = _oracle_c1(wf, head.weights, op.HyperparamsOracle(N=2048))
    ac = audio_complete(wf, p.N)
    assert ac.shape == (1025, 259)
    assert np.abs(ac.mag - mag_ref).max() / mag_ref.max() < 1e-4
    C = ac.slice_C(0.5, 1.0, p.pitch_frames, bins_per_tone=p.pitch_bins_per_tone)
    assert C.shape == (174, 8)
    assert np.abs(C - C_ref).max() / C_ref.max() < 1e-4
    for mode in (0, 1, 2):                             # f32 MFMA, split-bf16, split-fp16 (the default)
        head.set_mode(mode)
        y = head.classify(C / C.max())
        assert y.shape == (1, 1)
        assert abs(float(y[0, 0]) - float(y_ref[0, 0])) / float(y_ref[0, 0]) < 1e-4
        if abs(float(y_ref[0, 0]) - np.floor(float(y_ref[0, 0])) - 0.5) > 1e-2:
            assert np.rint(y[0, 0]) == np.rint(y_ref[0, 0])
