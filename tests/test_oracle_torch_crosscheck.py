"""CPU: the RDCNN oracle (oracle/rdcnn.py, numpy) against PyTorch's own fp32 operators.

The reference ships neither weights nor recorded network outputs, so the oracle cannot be
pinned against Keras itself.  What can be checked independently is that its building
blocks compute what the framework documentation says they compute: this file re-evaluates the
same graph with torch.nn.functional (a third-party implementation of conv / batch-norm /
pooling / linear) and requires agreement to float32 rounding.  The asymmetric "same" padding
of even kernels (TensorFlow: pad_before = (k-1)//2, the rest after) is applied explicitly on
the torch side, as TF documents it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rdcnn as orc


def _t(x):            # NHWC numpy -> NCHW torch
    return torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2).contiguous()


def _n(x):            # NCHW torch -> NHWC numpy
    return x.permute(0, 2, 3, 1).contiguous().numpy()


def _conv_same_torch(x, k, b):
    kh, kw = k.shape[:2]
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = F.pad(_t(x), (pl, kw - 1 - pl, pt, kh - 1 - pt))
    w = torch.from_numpy(np.ascontiguousarray(k)).permute(3, 2, 0, 1).contiguous()
    return _n(F.conv2d(xp, w, torch.from_numpy(b)))


@pytest.mark.parametrize('kh,kw,cin,cout,H,W', [(4, 16, 1, 32, 20, 37), (4, 16, 32, 32, 7, 21),
                                                  (4, 2, 32, 64, 11, 8), (2, 2, 64, 64, 5, 3),
                                                  (3, 3, 8, 8, 6, 6)])
def test_conv_same(kh, kw, cin, cout, H, W):
    rng = np.random.default_rng(kh * 100 + kw)
    x = rng.standard_normal((2, H, W, cin)).astype(np.float32)
    k = (rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got, want = orc.conv2d_same(x, k, b), _conv_same_torch(x, k, b)
    assert got.shape == want.shape == (2, H, W, cout)
    assert np.abs(got - want).max() < 2e-5 * max(np.abs(want).max(), 1.0)
    # an impulse shows the padding split directly: output (r, c) = k[pt + r0 - r, pl + c0 - c]
    imp = np.zeros((1, H, W, 1), np.float32)
    r0, c0 = H // 2, W // 2
    imp[0, r0, c0, 0] = 1.0
    k1 = rng.standard_normal((kh, kw, 1, 1)).astype(np.float32)
    y = orc.conv2d_same(imp, k1, np.zeros(1, np.float32))[0, :, :, 0]
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    for dy in range(kh):
        for dx in range(kw):
            r, c = r0 + pt - dy, c0 + pl - dx
            if 0 <= r < H and 0 <= c < W:
                assert y[r, c] == k1[dy, dx, 0, 0]


def test_batchnorm_pool_dense():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 10, 9, 16)).astype(np.float32)
    w = {'p/gamma': rng.uniform(0.5, 2, 16).astype(np.float32), 'p/beta': rng.standard_normal(16).astype(np.float32),
         'p/mean': rng.standard_normal(16).astype(np.float32), 'p/var': rng.uniform(0.2, 3, 16).astype(np.float32)}
    want = _n(F.batch_norm(_t(x), torch.from_numpy(w['p/mean']), torch.from_numpy(w['p/var']),
                           torch.from_numpy(w['p/gamma']), torch.from_numpy(w['p/beta']), False, 0.0, 1e-3))
    assert np.abs(orc.batchnorm(x, w, 'p') - want).max() < 1e-5
    for pool in ((2, 2), (2, 8), (3, 1)):
        assert np.array_equal(orc.pool2d(x, pool, 'max'), _n(F.max_pool2d(_t(x), pool)))
        assert np.abs(orc.pool2d(x, pool, 'avg') - _n(F.avg_pool2d(_t(x), pool))).max() < 1e-6


def _forward_torch(w, cfg, xs, dtype=torch.float32):
    """The graph of RDCNN.py:176-233 written with torch.nn.functional (any tower count, the
    1x1-conv + avg-pool + BN shortcut projection of RDCNN.py:312-335 included)."""
    ndt = np.float32 if dtype == torch.float32 else np.float64
    tw = {k: torch.from_numpy(np.asarray(v, ndt)) for k, v in w.items()}

    def bn(x, p):
        return F.batch_norm(x, tw[p + '/mean'], tw[p + '/var'], tw[p + '/gamma'], tw[p + '/beta'], False, 0.0, 1e-3)

    flats = []
    rfreq = cfg['residual_layer_frequencies']
    n_proj = 0
    for t, x in enumerate(xs):
        p1 = _t(np.asarray(x, ndt))
        p0 = [p1] * len(rfreq)
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            k = tw['t%d/conv%d/kernel' % (t, i)]
            kh, kw = k.shape[:2]
            pt, pl = (kh - 1) // 2, (kw - 1) // 2
            p1 = F.conv2d(F.pad(p1, (pl, kw - 1 - pl, pt, kh - 1 - pt)), k.permute(3, 2, 0, 1).contiguous(),
                          tw['t%d/conv%d/bias' % (t, i)])
            p1 = torch.sigmoid(bn(p1, 't%d/bn%d' % (t, i)))
            for ri in range(len(rfreq)):
                if i % rfreq[ri] == 0:
                    a = p0[ri]
                    if a.shape != p1.shape:
                        n_proj += 1
                        if a.shape[1] != p1.shape[1]:
                            sk = tw['t%d/sc%d/kernel' % (t, i)]
                            a = F.conv2d(a, sk.permute(3, 2, 0, 1).contiguous(), tw['t%d/sc%d/bias' % (t, i)])
                        if a.shape[2:] != p1.shape[2:]:
                            st = (a.shape[2] // p1.shape[2], a.shape[3] // p1.shape[3])
                            a = F.avg_pool2d(a, st)
                        a = bn(a, 't%d/scbn%d' % (t, i))
                    p1 = bn(a + p1, 't%d/resbn%d' % (t, i))
                    p0[ri] = p1
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1 = F.max_pool2d(p1, tuple(cfg['pool_sizes'][t]))
        flats.append(p1.permute(0, 2, 3, 1).reshape(p1.shape[0], -1))      # Keras flatten: (H, W, C)
    c = torch.cat(flats, 1)
    m = torch.sigmoid(c @ tw['dense1/kernel'] + tw['dense1/bias'])
    lg = m @ tw['dense2/kernel'] + tw['dense2/bias']
    if cfg['output_classes'] > 1:
        y = torch.softmax(lg, dim=1)
    else:
        lo, hi = cfg['output_range']
        y = torch.sigmoid(lg) * (hi - lo) + lo
    return lg.numpy(), y.numpy(), n_proj


def _head(name, n_fft):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'amt-saga_amd'))
    from amt_saga import heads, hyperparams
    p = hyperparams.Hyperparams(N=n_fft)
    if name == 'velocity':
        h = heads.VelocityClassifier(p)
    elif name == 'pitch':
        h = heads.pitch_classifier(p)
    elif name == 'timing':
        h = heads.timming_classifier(p)
    else:
        h = heads.InstrumentClassifier(p, name)
    return p, h, orc.head_config(p, name)


@pytest.mark.parametrize('name,n_fft,B', [('velocity', 2048, 2), ('pitch', 2048, 2), ('instrument', 2048, 2),
                                          ('instrument_dual', 2048, 2), ('timing', 4096, 2), ('timing', 2048, 1)])
def test_full_forward_every_head(name, n_fft, B):
    """Every head graph the path uses -- pitch, velocity, instrument, the two-tower instrument_dual and the
    timing head at N = 4096 (20 x 258) and N = 2048 (20 x 516; > 97 % of the loop's flops: 4 x 16 kernels,
    2 x 8 pools, the 1x1-conv + avg-pool shortcut projections at layers 14 and 26) -- evaluated by torch's own
    conv / batch-norm / pooling / linear in float32 AND float64, against the numpy oracle in the same
    precision: logits, the output activation (softmax / sigmoid + range scaling, RDCNN.py:218-221,308-310)
    and the float64 pair to 1e-9."""
    p, h, cfg = _head(name, n_fft)
    rng = np.random.default_rng(3)
    xs = [(rng.random((B,) + tuple(s[:2]) + (1,)) ** 2).astype(np.float32) for s in cfg['input_shapes']]
    got_lg = orc.forward(h.weights, cfg, xs, np.float32, return_logits=True)
    got_y = orc.forward(h.weights, cfg, xs, np.float32)
    lg32, y32, n_proj = _forward_torch(h.weights, cfg, xs, torch.float32)
    assert got_lg.shape == lg32.shape == (B, cfg['output_classes'])
    assert n_proj == 3 * len(xs)                                  # input -> 32, 32 -> 64 + pool, 64 -> 128 + pool
    assert np.abs(got_lg - lg32).max() < 1e-4 * max(np.abs(lg32).max(), 1.0)
    assert np.abs(got_y - y32).max() < 1e-4 * max(np.abs(y32).max(), 1.0)
    got64 = orc.forward(h.weights, cfg, xs, np.float64, return_logits=True)
    goty64 = orc.forward(h.weights, cfg, xs, np.float64)
    lg64, y64, _ = _forward_torch(h.weights, cfg, xs, torch.float64)
    assert np.abs(got64 - lg64).max() < 1e-9 * max(np.abs(lg64).max(), 1.0)
    assert np.abs(goty64 - y64).max() < 1e-9 * max(np.abs(y64).max(), 1.0)
    # and the float32 evaluations are both float32-close to the float64 truth
    assert np.abs(got_lg - lg64).max() < 1e-4 * max(np.abs(lg64).max(), 1.0)


def test_oracle_stft_istft_vs_scipy():
    """oracle.audio.stft/istft (librosa semantics restated) against scipy.signal: centred frames
    with reflect ('even') extension, periodic Hann, no scaling; scipy divides by sum(window)."""
    from scipy import signal
    from oracle import audio as oa
    rng = np.random.default_rng(11)
    for n_fft, hop, L in ((512, 128, 128 * 40), (2048, 512, 512 * 21), (1024, 256, 256 * 33 + 77)):
        x = rng.standard_normal(L).astype(np.float32)
        D = oa.stft(x, n_fft, hop)
        win = signal.get_window('hann', n_fft, fftbins=True)
        _, _, Z = signal.stft(x.astype(np.float64), window=win, nperseg=n_fft, noverlap=n_fft - hop,
                              boundary='even', padded=False, return_onesided=True)
        Z = Z * win.sum()
        T = min(D.shape[1], Z.shape[1])
        assert abs(D.shape[1] - Z.shape[1]) <= 1          # scipy drops a trailing partial frame
        assert np.abs(D[:, :T] - Z[:, :T]).max() < 2e-4 * np.abs(Z).max()
        # inverse: both recover the signal (COLA)
        y = oa.istft(D, hop)
        n = min(len(y), L)
        assert np.abs(y[:n] - x[:n]).max() < 1e-4


def test_oracle_cqt_known_answers():
    """oracle.cqt closed-form checks: a stationary sinusoid at a bin's (quantised) centre frequency
    gives sqrt(N_k) * A / 2 (L1-normalised Hann, scale = True), independent of the frame position and
    of the phase; the response is linear in amplitude and falls off over neighbouring bins."""
    from oracle import cqt as ocqt
    sr, hop = 44100, 512
    bpo = 24
    inc, length, freq = ocqt.cqt_table(sr, 27.5 * 2 ** 3, 48, bpo)          # two octaves from A3
    L = hop * 120
    n = np.arange(L)
    for k in (5, 20, 40):
        fq = float(inc[k]) / 2.0 ** 32 * sr                                  # the quantised frequency
        for A, ph0 in ((0.7, 0.3), (0.2, 2.0)):
            x = A * np.cos(2 * np.pi * fq * n / sr + ph0)
            C = ocqt.cqt_frames(x, [40, 41, 57, 80], inc, length, hop)
            want = np.sqrt(length[k]) * A / 2.0
            assert np.abs(C[k] - want).max() / want < 2e-3, (k, C[k], want)   # image term ~ 1/(4 pi Q)
            assert C[k].std() / want < 1e-3                                   # frame-position independent
            assert np.argmax(C[:, 0]) == k
            assert C[k + 4, 0] < 0.05 * C[k, 0] and C[k - 4, 0] < 0.05 * C[k, 0]
    # frames outside the signal (zero padding) and the -1 marker
    x = np.cos(2 * np.pi * float(freq[10]) * n / sr)
    C = ocqt.cqt_frames(x, [-1, 0, 60], inc, length, hop)
    assert np.all(C[:, 0] == 0)
    assert C[10, 1] < 0.75 * C[10, 2]                                         # half of the first frame is padding


def test_oracle_cqt_complex_known_answers():
    """Complex form (util_audio.py:428, magnitude_only=False): |C| equals the magnitude form exactly; for a stationary
    cosine A cos(2 pi f n / sr + ph0) at a bin's quantised centre frequency the value referred to the frame's centre
    is (sqrt(N_k) A / 2) e^{+i (ph0 + 2 pi f t hop / sr)} -- the signal's own phase at the frame centre -- up to the
    image term; and it is the direct centred-filter sum sum_n x[t hop - N_k//2 + n] w[n] e^{-2 pi i f (n - N_k//2) / sr}."""
    from oracle import cqt as ocqt
    sr, hop, bpo = 44100, 512, 24
    inc, length, freq = ocqt.cqt_table(sr, 220.0, 48, bpo)
    L = hop * 120
    n = np.arange(L)
    frames = [40, 41, 57, 80, -1]
    rng = np.random.default_rng(0)
    noise = rng.standard_normal(L)
    Cn = ocqt.cqt_frames(noise, frames, inc, length, hop, complex_out=True)
    assert np.abs(np.abs(Cn) - ocqt.cqt_frames(noise, frames, inc, length, hop)).max() < 1e-12
    assert np.all(Cn[:, 4] == 0)
    for k in (5, 20, 40):
        fq = float(inc[k]) / 2.0 ** 32                                       # cycles per sample, quantised
        A, ph0 = 0.7, 0.3
        x = A * np.cos(2 * np.pi * fq * n + ph0)
        C = ocqt.cqt_frames(x, frames[:4], inc, length, hop, complex_out=True)
        for j, t in enumerate(frames[:4]):
            want = np.sqrt(length[k]) * A / 2.0 * np.exp(1j * (ph0 + 2 * np.pi * fq * t * hop))
            assert abs(C[k, j] - want) / abs(want) < 2e-3, (k, t, C[k, j], want)
        nk = int(length[k])
        a = frames[0] * hop - nk // 2
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nk) / nk)
        direct = np.sum(noise[a:a + nk] * w * np.exp(-2j * np.pi * fq * (np.arange(nk) - nk // 2))) * 2 / np.sqrt(nk)
        assert abs(Cn[k, 0] - direct) < 1e-9 * max(abs(direct), 1.0)


def test_oracle_cqt_window_max_equals_all_frames():
    """ref_C_* = max of the whole CQT (training.py:271-282).  oracle.cqt.cqt_window_max evaluates it in O(L) per
    bin from cumulative sums; here it is pinned to the definition -- cqt_frames on every frame 0 .. L // hop -- for
    filters shorter than a hop, longer than the window and 30 x longer than the window."""
    from oracle import cqt as ocqt
    sr, hop = 44100, 512
    rng = np.random.default_rng(1)
    L = hop * 40 + 123
    t = np.arange(L) / sr
    x = np.sin(2 * np.pi * 440 * t) * np.exp(-3 * t) + 0.3 * np.sin(2 * np.pi * 1000 * t + 1) + 0.05 * rng.standard_normal(L)
    for fmin, n_bins, bpo in ((27.5, 87, 12), (2000.0, 24, 24), (27.5, 12, 192)):
        inc, length, _ = ocqt.cqt_table(sr, fmin, n_bins, bpo)
        brute = ocqt.cqt_frames(x, np.arange(1 + L // hop), inc, length, hop).max()
        fast = ocqt.cqt_window_max(x, inc, length, hop)
        assert abs(brute - fast) <= 1e-12 * brute, (fmin, brute, fast)
    assert ocqt.cqt_window_max(np.zeros(L), inc, length, hop) == 0.0


def test_cqt_spec_vs_librosa_like_algorithm():
    """How far is the build's CQT specification (oracle/cqt.py: direct transform, zero padding) from what
    librosa's recursive algorithm computes?  oracle/cqt_librosa_like.py restates that algorithm (octave-wise
    decimation, sparsified FFT basis, reflect padding; scipy's polyphase resampler stands in for resampy) and
    the two are compared on a synthetic three-note window for the parameterisations slice_C is called with
    (training.py:340-388): interior frames agree to < 1 % of the spectrogram's maximum (measured 0.05 - 0.3 %);
    frames within a filter length of the window's ends differ by the padding rule (reflect vs zero)."""
    from oracle import cqt as ocqt, cqt_librosa_like as ll, synth as osynth
    sr, hop = 44100, 512
    L = hop * 200
    x = osynth.render_window([(0, 57, 100, 0.3, 1.0), (0, 64, 90, 0.8, 0.8), (0, 45, 110, 1.2, 0.6)], L, sr).numpy()
    report = {}
    for name, fmin_midi, n_bins, bpo in (('pitch', 21, 174, 24), ('instrument', 21, 348, 48), ('velocity', 47, 36, 24)):
        fmin = 440.0 * 2 ** ((fmin_midi - 69) / 12)
        Cl = ll.cqt_mag(x, sr, hop, fmin, n_bins, bpo)
        inc, length, _ = ocqt.cqt_table(sr, fmin, n_bins, bpo)
        frames = np.arange(0, 200, 7)
        Cd = ocqt.cqt_frames(x, frames, inc, length, hop)
        A = Cl[:, frames]
        # interior entries: the bin's filter, centred on the frame, lies inside the window (no padding involved)
        c = frames[None, :] * hop
        half = (length[:, None] // 2) + hop
        interior = (c - half >= 0) & (c + half <= L)
        assert interior.any(axis=1).mean() > 0.6, name            # most bins have interior frames in 2.3 s
        dev = np.abs(A - Cd)[interior].max() / Cd.max()
        report[name] = dev
        assert dev < 0.01, (name, dev)
        assert np.corrcoef(A[interior], Cd[interior])[0, 1] > 0.9995
        # outside, the padding rule shows: librosa reflects, the build pads with zeros
        report[name + ' (edge frames)'] = np.abs(A - Cd)[~interior].max() / Cd.max() if (~interior).any() else 0.0
    print('CQT spec vs librosa-like algorithm, max deviation / max:', {k: round(float(v), 5) for k, v in report.items()})
