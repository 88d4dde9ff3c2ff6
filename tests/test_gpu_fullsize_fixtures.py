"""GPU parity at the metric size against fixtures oracle/ computed in the build container
(tests/golden/gen_fullsize_fixtures.py -> tests/golden/fullsize_fixtures.npz): config C3 on 16 windows of 516 frames,
config C5 as stated (all heads, five iterations, instrument groups 0-2) on 3, and the song-level normalisers of every
one of those windows on all three CQT grids (87 / 348 / 1392 bins, training.py:271-282).

The oracle ran on its OWN normalisers (LoopOracle.ref_levels), not on the product's; the product runs on its own
(prepare()), and the two sets are compared to 1e-4 for every window.  Events bit for bit; the heads' pre-rounding
floats within the bands of oracle/compare.py; the residual to 1e-4 of its maximum -- in full (16-bit quantised, step
1.5e-5 of the maximum) for the first windows of a case, as per-frame maxima and 20-band compression for all."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import fixture_waves as fw                                   # noqa: E402

REF_KEYS = ('ref_mag', 'ref_C_1', 'ref_C_inst', 'ref_C_foc')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(HERE, 'golden', 'fullsize_fixtures.npz'))


def _waves(case, fx):
    idx = fx[case + '_idx']
    notes = fw.note_lists(case, int(idx.max()) + 1)
    pcm = fw.render_pcm(case, [notes[i] for i in idx])
    assert fw.sha1(pcm) == str(fx[case + '_sha1']), 'the re-rendered 24-bit audio differs from the fixture\'s input'
    return fw.pcm_to_wave(pcm)


@pytest.mark.parametrize('case,streams', [('c3', 1), ('c5', 1), ('c3', 2), ('c5', 2)])
def test_fullsize_loop_vs_oracle_fixture(case, streams, fx):
    import torch
    from amt_saga.loop import TranscriptionLoop
    from oracle import audio as oa
    from oracle.compare import bands_for
    c = fw.CASES[case]
    p = fw.params_for(case)
    wave_h = _waves(case, fx)
    B = wave_h.shape[0]
    lp = TranscriptionLoop(p, heads=c['heads'], iters=c['iters'], groups=c['groups']).setup_device()
    lp.timing_streams = streams                               # 2: timing_end on a second stream under timing_start (opt-in)
    lp.trace = []
    events, b = lp.run(torch.from_numpy(wave_h).cuda())
    torch.cuda.synchronize()
    trace, lp.trace = [{k: v.cpu().numpy() for k, v in t.items()} for t in lp.trace], None
    # the product's own song-level normalisers against the oracle's own, every window, every grid the case uses
    want_refs = fx[case + '_refs']
    for j, k in enumerate(REF_KEYS):
        if k in lp.refs:
            got = lp.refs[k].cpu().numpy()
            rel = np.abs(got - want_refs[:, j]) / want_refs[:, j]
            assert rel.max() < 1e-4, (case, k, rel)
        else:
            assert np.all(np.isnan(want_refs[:, j])), k
    ev = events.cpu().numpy()
    assert np.array_equal(ev, fx[case + '_events']), (case, np.argwhere(ev != fx[case + '_events']))
    bands = bands_for(p)
    worst = {}
    for it in range(c['iters']):
        for name, g in trace[it].items():
            want = fx[case + '_float_' + name][it]
            d = np.abs(g.reshape(B, -1).astype(np.float64) - want.reshape(B, -1)).max()
            worst[name] = max(worst.get(name, 0.0), float(d))
            assert d <= bands[name], (case, name, it, d, bands[name])
    mag = b.mag.cpu().numpy()                                 # [B, T, ldf] frame-major
    F = p.N // 2 + 1
    rmax = b.ref_max.cpu().numpy()
    for i in range(B):
        m = mag[i][:, :F].T
        scale = float(fx[case + '_fmax'][i].max())
        assert np.abs(m.max(axis=0) - fx[case + '_fmax'][i]).max() <= 1e-4 * scale, ('frame maxima', case, i)
        assert abs(float(rmax[i]) - scale) <= 1e-4 * scale, ('ref_max', case, i)
        band = oa.AudioCompleteOracle.compress_bands(m, bands=p.timing_bands)
        assert np.abs(band - fx[case + '_band'][i]).max() <= 1e-4 * float(fx[case + '_band'][i].max()), ('bands', case, i)
    for i, s in enumerate(fx[case + '_resid_scale']):
        want = fx[case + '_resid_q'][i].astype(np.float64) * s
        m = mag[i][:, :F].T
        assert np.abs(m - want).max() <= (1e-4 + 1.0 / 65535) * want.max(), ('residual', case, i)
    print('%s (%d timing stream%s): %d windows of %d frames x %d iterations against the oracle fixture: events bit-exact, '
          'worst float distances %s' % (case, streams, 's' * (streams > 1), B, p.timing_frames, c['iters'],
                                        {k: '%.2g' % v for k, v in worst.items()}))


def test_fullsize_normaliser_grids_both_forms(fx):
    """cqt_max_mfma_kernel AND the VALU form at 516 frames on the 87- / 348- / 1392-bin grids against the oracle's
    maxima of C5's three windows (the largest grid had no oracle comparison at this size before round 4)."""
    import torch
    from amt_saga import audio
    p = fw.params_for('c5')
    wave = torch.from_numpy(_waves('c5', fx)).cuda()
    f_lo = float(audio.midi_to_hz(p.pitch_low))
    span = p.pitch_high - p.pitch_low
    want = fx['c5_refs']
    for j, mult in ((1, 1), (2, p.instrument_bins_per_tone), (3, 4 * p.instrument_bins_per_tone)):
        tb = audio.cqt_table(p.sr, f_lo, span * mult, 12 * mult, 'cuda')
        for form in ('mfma', 'valu'):
            got = audio.cqt_window_max(wave, tb, p.H, form=form).cpu().numpy()
            assert (np.abs(got - want[:, j]) / want[:, j]).max() < 1e-4, (mult, form, got, want[:, j])
