"""GPU parity: HIP STFT / magphase / iSTFT / subtract / features vs the oracle,
the reference-emitted golden vectors and the recorded FLAC triples.  All calls
go through the C ABI (ctypes) -- no CPU fallback exists in the product path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4      # north_star tolerance: relative to the window maximum


def _relmax(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(
        np.abs(b).max(), 1e-30)


@pytest.fixture(scope='module')
def mods():
    import torch
    assert torch.cuda.is_available()
    from amt_saga import audio, _lib
    _lib.load()
    from oracle import audio as oa
    return audio, oa


def _signal(L, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / 44100.0
    y = 0.5 * np.sin(2 * np.pi * 220.0 * t) + 0.3 * np.sin(2 * np.pi * 1333.7 * t + 1.0)
    y += 0.05 * rng.standard_normal(L)
    y *= np.exp(-t * 1.5)
    return y.astype(np.float32)


@pytest.mark.parametrize('n_fft,L', [(256, 1000), (512, 4096), (1024, 5000), (2048, 263680 // 8),
                                     (2048, 20000), (4096, 4096 * 6 + 17), (4096, 132300)])
def test_stft_mag_phase_max(mods, n_fft, L):
    audio, oa = mods
    B = 3
    wave = np.stack([_signal(L, s) for s in range(B)])
    b = audio.AudioBatch(wave, n_fft).stft(True)
    mag = b.mag.cpu().numpy()
    ph = b.ph.cpu().numpy()
    rmax = b.ref_max.cpu().numpy()
    Fb = n_fft // 2 + 1
    for i in range(B):
        F = oa.stft(wave[i], n_fft)
        m_ref, p_ref = oa.magphase(F)
        assert mag[i].shape[0] == F.shape[1]
        assert _relmax(mag[i][:, :Fb].T, m_ref) < REL
        assert np.all(mag[i][:, Fb:] == 0)
        # phase: compare where the magnitude is not negligible (angle is ill-conditioned at 0)
        sel = m_ref > 1e-3 * m_ref.max()
        pc = (ph[i][:, :Fb, 0] + 1j * ph[i][:, :Fb, 1]).T
        assert np.abs(pc - p_ref)[sel].max() < 2e-3
        assert np.abs(np.abs(pc) - 1).max() < 1e-5
        assert abs(rmax[i] - mag[i].max()) == 0
        assert abs(rmax[i] - m_ref.max()) / m_ref.max() < REL


def test_stft_edge_cases(mods):
    audio, oa = mods
    # silence -> mag 0, phase 1+0i (magphase of 0), max 0
    b = audio.AudioBatch(np.zeros((1, 3000), np.float32), 512).stft(True)
    assert float(b.mag.abs().max()) == 0.0
    ph = b.ph.cpu().numpy()[0][:, :257]
    assert np.all(ph[..., 0] == 1) and np.all(ph[..., 1] == 0)
    assert float(b.ref_max[0]) == 0.0
    # too short for the reflect padding -> ValueError (shape), like numpy.pad would refuse
    with pytest.raises(ValueError):
        audio.AudioBatch(np.zeros((1, 100), np.float32), 512).stft()
    # unsupported FFT size
    with pytest.raises(ValueError):
        audio.AudioBatch(np.zeros((1, 3000), np.float32), 300)


@pytest.mark.parametrize('n_fft,T', [(512, 37), (1024, 33), (2048, 64), (2048, 65), (4096, 130)])
def test_istft_vs_oracle_and_roundtrip(mods, n_fft, T):
    audio, oa = mods
    hop = n_fft // 4
    L = hop * (T - 1)
    wave = np.stack([_signal(L, 10 + s) for s in range(2)])
    b = audio.AudioBatch(wave, n_fft).stft(True)
    assert b.T == T
    y = b.istft().cpu().numpy()
    assert y.shape == (2, L)
    # STFT -> iSTFT is the identity (COLA, hann, hop = N/4)
    assert _relmax(y, wave) < REL
    # modified spectrogram (not a consistent STFT): compare with the oracle's istft
    import torch
    b.mag *= torch.linspace(0.2, 1.0, b.mag.shape[2], device=b.mag.device)
    y2 = b.istft().cpu().numpy()
    Fb = n_fft // 2 + 1
    for i in range(2):
        m = b.mag[i].cpu().numpy()[:, :Fb].T
        p = b.ph[i].cpu().numpy()
        pc = (p[:, :Fb, 0] + 1j * p[:, :Fb, 1]).T
        ref = oa.istft(m * pc, hop)
        assert _relmax(y2[i], ref) < REL


def test_istft_full_size_batch(mods):
    """BASELINE-sized windows (516 frames) at a batch large enough for the 16-hop segments of the
    streaming iSTFT kernel: STFT -> iSTFT is the identity for every window; a modified (inconsistent)
    spectrogram matches the oracle's istft; a hop != N/4 plan (generic kernel) agrees as well."""
    audio, oa = mods
    import torch
    n_fft, hop, T, B = 2048, 512, 516, 66
    L = hop * (T - 1)
    base = np.stack([_signal(L, 40 + s) for s in range(3)])
    wave = np.concatenate([base] * (B // 3))
    b = audio.AudioBatch(wave, n_fft).stft(True)
    y = b.istft().cpu().numpy()
    assert _relmax(y, wave) < REL
    b.mag *= torch.linspace(1.0, 0.1, b.mag.shape[2], device=b.mag.device)
    y2 = b.istft().cpu().numpy()
    Fb = n_fft // 2 + 1
    for i in (0, B - 1):
        m = b.mag[i].cpu().numpy()[:, :Fb].T
        p = b.ph[i].cpu().numpy()
        pc = (p[:, :Fb, 0] + 1j * p[:, :Fb, 1]).T
        assert _relmax(y2[i], oa.istft(m * pc, hop)) < REL
    assert np.array_equal(y2[0], y2[3])                       # same window, other place in the batch
    # hop = N/8: the generic gather kernel
    hop8 = 256
    L8 = hop8 * 80
    w8 = np.stack([_signal(L8, 50 + s) for s in range(2)])
    b8 = audio.AudioBatch(w8, n_fft, hop8).stft(True)
    y8 = b8.istft().cpu().numpy()
    assert _relmax(y8, w8) < REL


def test_subtract_golden_bit_exact(mods, refvec):
    """amt_subtract vs the reference's own subtract() outputs: bit-exact."""
    audio, oa = mods
    n = int(refvec['sub_cases'])
    for c in range(n):
        n_fft, off_s, acomp, norm, relu, overkill = refvec['sub%d_args' % c]
        n_fft = int(n_fft)
        Tm = refvec['sub%d_mix' % c].shape[1]
        hop = n_fft // 4
        ac = audio.audio_complete(np.zeros(hop * (Tm - 1), np.float32), n_fft)
        ac._mag = refvec['sub%d_mix' % c].copy()
        g = audio.audio_complete(np.zeros(hop * (refvec['sub%d_guess' % c].shape[1] - 1), np.float32), n_fft)
        g._mag = refvec['sub%d_guess' % c].copy()
        ac.subtract(g, offset=float(off_s), attack_compensation=int(acomp),
                    normalize=bool(norm), relu=bool(relu), overkill_factor=float(overkill))
        out = ac.mag
        exp = refvec['sub%d_out' % c]
        assert out.dtype == np.float32 and exp.dtype == np.float32
        assert np.array_equal(out, exp), 'case %d max diff %g' % (c, np.abs(out - exp).max())
        # invalidation state (util_audio.py:149-157)
        assert ac._wf is None and ac._F is None and ac._D is None and ac._ref_mag is None
        assert ac.ref_mag == exp.max()
    # raw ndarray subtrahend
    ac = audio.audio_complete(np.zeros(512 * 23, np.float32), 2048)
    ac._mag = refvec['subraw_mix'].copy()
    ac.subtract(refvec['subraw_guess'].copy(), offset=0.05)
    assert np.array_equal(ac.mag, refvec['subraw_out'])
    # offset beyond the window: the reference raises ValueError
    ac = audio.audio_complete(np.zeros(64 * 39, np.float32), 256)
    ac._mag = np.ones((129, 40), np.float32)
    with pytest.raises(ValueError):
        ac.subtract(np.ones((129, 13), np.float32), offset=10.0)


def test_subtract_batched_bank(mods):
    """Batched form: guess bank + per-window index / offset / length vs the oracle loop."""
    audio, oa = mods
    import torch
    rng = np.random.default_rng(5)
    n_fft, B, T, G, Tg = 512, 7, 50, 4, 12
    Fb, ldf = 257, audio.ldf_of(512)
    mix = (rng.random((B, Fb, T)) ** 2).astype(np.float32)
    bank = (rng.random((G, Fb, Tg)) ** 2).astype(np.float32)
    idx = rng.integers(0, G, B).astype(np.int32)
    off = np.array([0, 3, 38, 45, 49, 50, 10], np.int32)     # 49/50: clipped to <=1/0 frames
    gl = np.array([12, 5, 12, 12, 12, 12, 0], np.int32)      # ragged guess lengths incl. empty
    b = audio.AudioBatch(None, n_fft)
    hm = np.zeros((B, T, ldf), np.float32); hm[:, :, :Fb] = mix.transpose(0, 2, 1)
    hb = np.zeros((G, Tg, ldf), np.float32); hb[:, :, :Fb] = bank.transpose(0, 2, 1)
    b.mag = torch.from_numpy(hm).cuda()
    b.ref_max = b.window_max()
    assert np.array_equal(b.ref_max.cpu().numpy(), mix.max(axis=(1, 2)))
    gmax = torch.from_numpy(bank.max(axis=(1, 2))).cuda()
    b.subtract(torch.from_numpy(hb).cuda(), gmax, torch.from_numpy(idx).cuda(),
               torch.from_numpy(gl).cuda(), torch.from_numpy(off).cuda(), True, True, 1.0)
    out = b.mag.cpu().numpy()[:, :, :Fb].transpose(0, 2, 1)
    for i in range(B):
        exp = mix[i].copy()
        g = bank[idx[i]][:, :gl[i]].copy()
        g *= exp.max() / bank[idx[i]].max()
        n = max(0, min(gl[i], T - off[i]))
        exp[:, off[i]:off[i] + n] -= g[:, :n]
        exp = np.maximum(exp, 0)
        assert np.array_equal(out[i], exp), i
        assert float(b.ref_max[i]) == exp.max()


import recorded as rec      # noqa: E402  (tests/recorded.py)


@pytest.mark.parametrize('name', rec.frozen_triples())
def test_flac_triples_through_hip(mods, name):
    """The reference's recorded librosa outputs (subtraction_demo, 15 scenarios) through the HIP chain
    STFT -> subtract -> iSTFT of the drop-in audio_complete: within fp32 FFT error of the recording on every
    sample the recorded inputs determine."""
    audio, oa = mods
    y, sub, m, z = rec.run_triple(audio.audio_complete, name)
    assert y.shape == sub.shape and m.sum() >= 100000
    assert np.abs(y - sub)[m].max() < 2e-5


@pytest.mark.parametrize('prog', sorted(rec.index()['short_windows']))
def test_short_window_demo_through_hip(mods, prog):
    """short_window_demo recordings: STFT -> audio_complete.resize(j, ['mag','ph']) -> iSTFT on the device
    returns the first hop * (j - 1) samples of the recording it was cut from (see test_oracle_golden)."""
    audio, oa = mods
    d = rec.index()['short_windows'][prog]
    x20 = (rec.wave(d['20']) * rec.SCALE).astype(np.float32)
    for j in (6, 8, 10, 15, 20):
        ac = audio.audio_complete(x20, 4096)
        sw = ac.resize(0.0, len(x20) / 44100.0, j, attribs=['mag', 'ph'])
        assert sw.shape == (2049, j)
        y = sw.wf
        assert y.shape == (1024 * (j - 1),) == rec.wave(d[str(j)]).shape
        assert np.abs(y - x20[:len(y)]).max() < 2e-5


@pytest.mark.parametrize('base', sorted(rec.index()['window_dumps']))
def test_window_dumps_through_hip(mods, base):
    """The in-loop subtract(normalize=True) dumps of training.py:438-447 (258 frames, N = 4096) through the HIP
    chain: same loose bound as the oracle test, and equal to the oracle to 1e-4."""
    audio, oa = mods
    r = rec.index()['window_dumps'][base]
    fw, g, af = (rec.wave(r[k]) * rec.SCALE for k in ('full_window', 'guessed', 'after_subtr'))
    outs = []
    for AC in (audio.audio_complete, oa.AudioCompleteOracle):
        A = AC(fw.astype(np.float32), 4096)
        G = AC(g.astype(np.float32), 4096)
        A.subtract(G, offset=A._frames_to_seconds(r['onset_frame']) + 1e-6)
        outs.append(np.asarray(A.wf))
    rms = lambda v: float(np.sqrt(np.mean(v ** 2)))
    assert rms(outs[0] - af) < 0.02 * rms(af)
    assert np.abs(outs[0] - outs[1]).max() < 1e-4 * np.abs(outs[1]).max()


def test_compress_bands_and_short_window(mods, refvec):
    audio, oa = mods
    import torch
    for Fb in (1025, 2049):
        S = refvec['cb_in_%d' % Fb]
        out = audio.audio_complete.compress_bands(S, bands=20)
        assert out.shape == (20, S.shape[1])
        assert _relmax(out, refvec['cb_out_%d' % Fb]) < 1e-6
    out = audio.audio_complete.compress_bands(refvec['cb_lin_in'], bands=8, log=False)
    assert _relmax(out, refvec['cb_lin_out']) < 1e-6
    # batched C_timing with reference division and a resize table
    rng = np.random.default_rng(3)
    n_fft, B, T = 2048, 3, 21
    wave = np.stack([_signal(512 * (T - 1), 30 + s) for s in range(B)])
    b = audio.AudioBatch(wave, n_fft).stft(True)
    ref = b.ref_max.clone()
    ct = b.compress_bands(20, ref, 32).cpu().numpy()
    for i in range(B):
        m = b.mag[i].cpu().numpy()[:, :1025].T
        exp = oa.AudioCompleteOracle._resize(oa.AudioCompleteOracle.compress_bands(m, 20), 32) / m.max()
        assert _relmax(ct[i], exp) < 1e-5
    # short windows: three modes vs the oracle recipe (training.py:347-363)
    src = np.stack([audio.resize_source_frames(t, 8) + s for t, s in ((3, 2), (11, 5), (0, 0))]).astype(np.int32)
    src[2] = -1
    lo = np.array([11, 700, 40], np.int32)
    srcd, lod = torch.from_numpy(src).cuda(), torch.from_numpy(lo).cuda()
    f0 = b.short_window(srcd, lod, 348, ref, 0).cpu().numpy()
    f1 = b.short_window(srcd, lod, 348, None, 1).cpu().numpy()
    f2 = b.short_window(srcd, lod, 348, None, 2).cpu().numpy()
    for i in range(B):
        m = b.mag[i].cpu().numpy()[:, :1025].T
        p = b.ph[i].cpu().numpy()
        pc = (p[:, :1025, 0] + 1j * p[:, :1025, 1]).T
        cols = src[i]
        sw = np.where(cols[None, :] >= 0, m[:, np.maximum(cols, 0)], 0)
        swp = np.where(cols[None, :] >= 0, pc[:, np.maximum(cols, 0)], 0)
        band = np.zeros((348, 8), np.float32); bandp = np.zeros((348, 8), np.complex64)
        hi = min(1025, lo[i] + 348)
        band[:hi - lo[i]] = sw[lo[i]:hi]; bandp[:hi - lo[i]] = swp[lo[i]:hi]
        assert _relmax(f0[i], band / m.max()) < 1e-6
        lg = np.log10(band * 1000 + 1)
        if lg.max() > 0:
            assert _relmax(f1[i], lg / lg.max()) < 1e-5
        else:
            assert np.all(np.isnan(f1[i]))          # 0/0, as in the reference
        # the angle wraps at the negative real axis: compare modulo the period 2*pi/6.3
        d = np.abs(f2[i] - (np.angle(bandp) + 3.15) / 6.3)
        per = 2 * np.pi / 6.3
        assert np.minimum(d, np.abs(per - d)).max() < 2e-4


def test_audio_complete_surface(mods):
    """Property cache / invalidation behaves like the reference object."""
    audio, oa = mods
    wf = _signal(1024 * 40, 77)
    a = audio.audio_complete(wf, 4096)
    o = oa.AudioCompleteOracle(wf, 4096)
    assert a.shape == o.shape == (2049, 41)
    assert _relmax(a.mag, o.mag) < REL
    assert abs(a.ref_mag - o.ref_mag) / o.ref_mag < REL
    assert a._seconds_to_frames(0.5) == o._seconds_to_frames(0.5)
    assert a.midi_tone_to_FFT(60) == o.midi_tone_to_FFT(60) == 23
    assert _relmax(np.abs(a.F), np.abs(o.F)) < REL
    s = a.section(0.1, None, 20); so = o.section(0.1, None, 20)
    assert s.shape == so.shape and _relmax(s.mag, so.mag) < REL
    r = a.resize(0.2, 0.1, 8, attribs=['mag', 'ph']); ro = o.resize(0.2, 0.1, 8, attribs=['mag', 'ph'])
    assert r.shape == ro.shape == (2049, 8) and _relmax(r.mag, ro.mag) < REL
    c = a.clone(); c.mag = c.mag * 0.5
    assert c._wf is None and c._ref_mag is None and _relmax(c.wf, 0.5 * wf[:1024 * 40]) < 2e-4
    with pytest.raises(ValueError):
        a._P('nope')
    with pytest.raises(ValueError):
        a.resize(0, 0.1, 8, attribs=['zzz'])


def _mkp(audio, n_fft, mag, ph=None):
    """Product object with injected spectra, as gen_golden_from_reference.py built the reference's."""
    T = mag.shape[1]
    ac = audio.audio_complete(np.zeros((n_fft // 4) * (T - 1), dtype=np.float32), n_fft)
    ac._mag = mag.copy()
    if ph is not None:
        ac._ph = ph.copy()
    return ac


def test_window_management_vs_reference_vectors(mods, refvec):
    """section / slice / concat / resize / section_power of the drop-in audio_complete -- frame index maps
    executed by amt_gather_frames on the device -- against the vectors the reference's own methods emitted
    (util_audio.py:286-382, 469-507): values bit-exact (they are copies), waveform lengths equal."""
    audio, oa = mods
    ac = _mkp(audio, 512, refvec['sec_mag'], refvec['sec_ph'])
    sec = ac.section(0.2, None, 50)
    assert sec._d['mag'] is not None and sec._h['mag'] is None          # cut on the device, never fetched
    assert np.array_equal(sec.mag, refvec['sec_out_mag'])
    assert np.array_equal(sec.ph, refvec['sec_out_ph'])
    assert len(sec._wf) == int(refvec['sec_out_wf_len'])
    sec2 = ac.section(0.1, 0.4)
    assert np.array_equal(sec2.mag, refvec['sec2_out_mag'])
    assert len(sec2._wf) == int(refvec['sec2_out_wf_len'])
    ac2 = ac.clone()
    ac2.slice(10, 40)
    assert np.array_equal(ac2.mag, refvec['slice_out_mag'])
    assert len(ac2._wf) == int(refvec['slice_out_wf_len'])
    ac2.concat(sec2)
    assert np.array_equal(ac2.mag, refvec['concat_out_mag'])
    assert ac2.ph.shape == ac2.mag.shape
    assert len(ac2._wf) == int(refvec['concat_out_wf_len'])
    lo = int(refvec['secpow_lo'])
    for i in range(4):
        start, dur = refvec['rsz%d_args' % i]
        rs = ac.resize(float(start), float(dur), 8, attribs=['mag', 'ph'])
        assert np.array_equal(rs.mag, refvec['rsz%d_mag' % i])
        assert np.array_equal(rs.ph, refvec['rsz%d_ph' % i])
        assert np.array_equal(rs.section_power('mag', lo, lo + 348), refvec['rsz%d_secpow' % i])
        assert np.array_equal(rs.section_power('mag', 200, 548), refvec['rsz%d_secpow_hi' % i])
    # an attribute only one side has is dropped by concat; slicing keeps views consistent
    a = _mkp(audio, 512, refvec['sec_mag'])
    a.concat(sec2)
    assert a._has('mag') and not a._has('ph')
    with pytest.raises(ValueError):
        ac.resize(0, 0.1, 8, attribs=['nope'])


def test_db_and_flatness_on_device(mods):
    """audio_complete.D (amt_amplitude_to_db: librosa.amplitude_to_db(mag, ref=ref_mag), floor at max - 80 dB),
    its inverse through the D setter (amt_db_to_amplitude) and spectral_flatness (amt_spectral_flatness)
    against the oracle's restatement of the librosa formulas."""
    audio, oa = mods
    wf = _signal(512 * 60, 5)
    wf[512 * 30:] *= 1e-6                                    # a quiet half: exercises the -80 dB floor and amin
    a = audio.audio_complete(wf, 2048)
    o = oa.AudioCompleteOracle(wf, 2048)
    D, Do = a.D, o.D
    assert D.shape == Do.shape == (1025, 61)
    assert np.abs(D - Do).max() < 2e-3                       # dB; float32 log10 of float32 magnitudes
    assert D.max() <= 1e-4 and abs(D.min() + 80.0) < 1e-3 and (D <= -79.99).mean() > 0.2
    # a different reference level, and no floor, through the batched entry point
    import torch
    m = a._dev('mag')[None]
    ref = torch.tensor([0.37], device='cuda')
    got = audio.amplitude_to_db(m, ref, 1025, top_db=None)[0].cpu().numpy()[:, :1025].T
    want = oa.amplitude_to_db(a.mag, ref=0.37, top_db=None)      # same magnitudes: the formula itself
    assert np.abs(got - want).max() < 2e-4 and want.min() < -90
    # D -> mag on the device: only D and ph set (util_audio.py:143-145)
    b = audio.audio_complete(None, 2048)
    b.ph = o.ph
    b.D = Do
    b._ref_mag = o.ref_mag
    mag_back = b.mag
    keep = Do > -79.0
    assert np.abs(mag_back - o.mag)[keep].max() / o.mag.max() < 1e-5
    # spectral flatness: tone-like signal vs white noise (the reference's render sanity check, > 0.3 = noise)
    tone = _signal(512 * 60, 6)
    f_tone = audio.audio_complete(tone, 2048).spectral_flatness()
    fo_tone = oa.AudioCompleteOracle(tone, 2048).spectral_flatness()
    assert abs(f_tone - fo_tone) < 1e-4 * max(fo_tone, 1e-3)
    noise = np.random.default_rng(0).standard_normal(512 * 60).astype(np.float32)
    fn = audio.audio_complete(noise, 2048).spectral_flatness()
    fo = oa.AudioCompleteOracle(noise, 2048).spectral_flatness()
    assert abs(fn - fo) < 1e-4 and fn > 0.3 > f_tone


def test_subtract_span_equals_whole_window(mods):
    """amt_subtract_span (only the frames the guess covers, per-frame maxima from amt_compress_bands_fmax) against
    amt_subtract on the same non-negative spectrograms: residual and new maxima bit for bit, the per-frame maxima equal
    to the rows' maxima afterwards; guesses of per-window lengths incl. 0, offsets at 0 / inside / running over the end;
    a second step from the updated maxima; and the fallbacks (no maxima, relu off)."""
    audio, _ = mods
    import torch
    rng = np.random.default_rng(11)
    B, T, n_fft = 5, 40, 2048
    wave = np.stack([_signal(512 * (T - 1), 70 + s) for s in range(B)])
    gw = np.stack([_signal(512 * 12, 90 + s) * (0.5 + s) for s in range(3)])
    g = audio.AudioBatch(gw, n_fft).stft(False)
    gidx = torch.tensor([0, 2, 1, 1, 0], dtype=torch.int32).cuda()
    gfr = torch.tensor([13, 0, 7, 13, 5], dtype=torch.int32).cuda()
    for offs in ([0, 3, 35, 39, 12], [1, 0, 0, 30, 37]):
        off = torch.tensor(offs, dtype=torch.int32).cuda()
        a = audio.AudioBatch(wave, n_fft).stft(False)
        b = audio.AudioBatch(wave, n_fft).stft(False)
        ref = a.ref_max.clone()
        a.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True)
        b.compress_bands(20, ref, T, fmax=True)
        fm = b._fmax[0]
        assert torch.equal(fm, b.mag[:, :, :1025].amax(dim=2))
        b.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True, span=True)
        assert b._fmax is not None                                   # the span kernel ran and kept the maxima
        assert torch.equal(a.mag, b.mag) and torch.equal(a.ref_max, b.ref_max)
        assert torch.equal(b._fmax[0], b.mag[:, :, :1025].amax(dim=2))
        # a second step straight from the updated maxima
        a.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True)
        b.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True, span=True)
        assert torch.equal(a.mag, b.mag) and torch.equal(a.ref_max, b.ref_max)
    # without maxima, or without the ReLU, span=True is the whole-window kernel
    c = audio.AudioBatch(wave, n_fft).stft(False)
    c.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True, span=True)
    d = audio.AudioBatch(wave, n_fft).stft(False)
    d.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=True)
    assert torch.equal(c.mag, d.mag) and c._fmax is None
    e = audio.AudioBatch(wave, n_fft).stft(False)
    e.compress_bands(20, None, T, fmax=True)
    e.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=False, span=True)
    f = audio.AudioBatch(wave, n_fft).stft(False)
    f.subtract(g.mag, g.ref_max, gidx, gfr, off, normalize=True, relu=False)
    assert torch.equal(e.mag, f.mag) and torch.equal(e.ref_max, f.ref_max) and e._fmax is None
    from amt_saga import _lib
    assert _lib.load().amt_compress_bands_fmax(c.mag.data_ptr(), B, T, 1025, c.ldf, T * c.ldf, gidx.data_ptr(), 20, None,
                                               gidx.data_ptr(), c.mag.data_ptr(), T, c.mag.data_ptr(), None) == _lib.AMT_E_INVALID
