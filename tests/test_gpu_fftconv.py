"""GPU parity of the FFT-domain form of one 4 x 16 convolution layer (amt_fftconv.hip, conv mode 3 of the RDCNN) against
a float64 direct convolution + BN + sigmoid (+ shortcut + BN) in numpy -- the operator oracle/rdcnn.py applies layer by
layer -- and against the float32 numpy result: the FFT form must sit as close to the float64 truth as float32 does."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _direct(a, k, dtype):
    B, H, W, Ci = a.shape
    KH, KW, _, Co = k.shape
    pt, pl = (KH - 1) // 2, (KW - 1) // 2
    ap = np.zeros((B, H + KH - 1, W + KW - 1, Ci), dtype)
    ap[:, pt:pt + H, pl:pl + W] = a
    y = np.zeros((B, H, W, Co), dtype)
    kk = k.astype(dtype)
    for dy in range(KH):
        for dx in range(KW):
            y += ap[:, dy:dy + H, dx:dx + W] @ kk[dy, dx]
    return y


def _layer(a, k, s1, t1, sc, s2, t2, dtype):
    z = _direct(a.astype(dtype), k, dtype) * s1.astype(dtype) + t1.astype(dtype)
    v = 1.0 / (1.0 + np.exp(-z))
    if sc is not None:
        v = (v + sc.astype(dtype)) * s2.astype(dtype) + t2.astype(dtype)
    return v


@pytest.mark.parametrize('B,H,W,residual', [(3, 20, 516, True), (2, 20, 258, False), (9, 5, 40, True), (1, 20, 561, False)])
def test_fftconv_layer_vs_numpy(B, H, W, residual):
    import torch
    from amt_saga import _lib
    lib = _lib.load()
    rng = np.random.default_rng(B * 1000 + W)
    a = rng.random((B, H, W, 32)).astype(np.float32)                     # sigmoid-range activations
    a[0] *= np.float32(3.0)                                              # windows of different scale: per-window operand scaling
    k = (rng.standard_normal((4, 16, 32, 32)) * 0.05).astype(np.float32)
    s1 = rng.uniform(0.5, 2.0, 32).astype(np.float32); t1 = rng.uniform(-1, 1, 32).astype(np.float32)
    s2 = rng.uniform(0.5, 2.0, 32).astype(np.float32); t2 = rng.uniform(-1, 1, 32).astype(np.float32)
    sc = rng.random((B, H, W, 32)).astype(np.float32) if residual else None
    ref64 = _layer(a, k, s1, t1, sc, s2, t2, np.float64)
    ref32 = _layer(a, k, s1, t1, sc, s2, t2, np.float32)
    h = C.c_void_p()
    fp = lambda x: x.ctypes.data_as(C.c_void_p)
    _lib.check(lib.amt_fftconv_create(C.byref(h), fp(k), fp(s1), fp(t1), fp(s2) if residual else None, fp(t2) if residual else None))
    try:
        ad = torch.from_numpy(a).cuda()
        scd = torch.from_numpy(sc).cuda() if residual else None
        out = torch.empty_like(ad)
        need = lib.amt_fftconv_workspace_bytes(B, H)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device='cuda')
        _lib.check(lib.amt_fftconv_run(h, ad.data_ptr(), scd.data_ptr() if residual else None, B, H, W, out.data_ptr(),
                                       ws.data_ptr(), need, 1, None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    finally:
        lib.amt_fftconv_destroy(h)
    scale = np.abs(ref64).max()
    e_gpu = np.abs(got - ref64).max() / scale
    e_cpu = np.abs(ref32 - ref64).max() / scale
    print('fftconv %dx%dx%d residual=%s: e_gpu %.3g  e_cpu(f32 direct) %.3g' % (B, H, W, residual, e_gpu, e_cpu))
    assert e_gpu < 1e-5, e_gpu
    assert e_gpu <= 2.5 * e_cpu + 2.4e-7, (e_gpu, e_cpu)


@pytest.mark.parametrize('B,residual,chain', [(3, True, 0), (9, False, 0), (2, True, 1), (70, False, 2)])
def test_fftpk_layer_vs_numpy(B, residual, chain):
    """The packed-image form of a 10 x 64, 64 -> 64 (4 x 16) layer (amt_fftpk.hip: the whole image as one 1152-point
    sequence per channel pair, one 128 x 128 GEMM per frequency pair) against the float64 direct convolution, under the
    bar of the other arithmetics; chain > 0: the layer applied chain + 1 times with the hand-over in the frequency domain
    (register epilogue, re-zeroed gaps), B = 70: more than one GEMM chunk, ragged."""
    import torch
    from amt_saga import _lib
    lib = _lib.load()
    H, W, Cn = 10, 64, 64
    rng = np.random.default_rng(B * 100 + chain)
    a = rng.random((B, H, W, Cn)).astype(np.float32)
    a[0] *= np.float32(3.0)
    k = (rng.standard_normal((4, 16, Cn, Cn)) * 0.04).astype(np.float32)
    s1 = rng.uniform(0.5, 2.0, Cn).astype(np.float32); t1 = rng.uniform(-1, 1, Cn).astype(np.float32)
    s2 = rng.uniform(0.5, 2.0, Cn).astype(np.float32); t2 = rng.uniform(-1, 1, Cn).astype(np.float32)
    sc = rng.random((B, H, W, Cn)).astype(np.float32) if residual else None

    def ref(dtype):
        v = a
        for _ in range(chain):
            v = _layer(v, k, s1, t1, None, s2, t2, dtype)
        return _layer(v, k, s1, t1, sc, s2, t2, dtype)
    ref64, ref32 = ref(np.float64), ref(np.float32)
    h = C.c_void_p()
    fp = lambda x: x.ctypes.data_as(C.c_void_p)
    _lib.check(lib.amt_fftpk_create(C.byref(h), fp(k), fp(s1), fp(t1), fp(s2) if residual else None, fp(t2) if residual else None))
    try:
        ad = torch.from_numpy(a).cuda()
        scd = torch.from_numpy(sc).cuda() if residual else None
        out = torch.empty_like(ad)
        need = lib.amt_fftpk_workspace_bytes(B)
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device='cuda')
        _lib.check(lib.amt_fftpk_run(h, ad.data_ptr(), scd.data_ptr() if residual else None, B, out.data_ptr(),
                                     ws.data_ptr(), need, chain, 1, None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
    finally:
        lib.amt_fftpk_destroy(h)
    scale = np.abs(ref64).max()
    e_gpu = np.abs(got - ref64).max() / scale
    e_cpu = np.abs(ref32 - ref64).max() / scale
    print('fftpk B=%d residual=%s chain=%d: e_gpu %.3g  e_cpu(f32 direct) %.3g' % (B, residual, chain, e_gpu, e_cpu))
    assert e_gpu < 1e-5, e_gpu
    assert e_gpu <= 2.5 * e_cpu + 2.4e-7, (e_gpu, e_cpu)


@pytest.mark.parametrize('form', ['row', 'packed'])
def test_fft_layer_beside_split_fp16_network_on_a_second_stream(form):
    """Regression for round 4's two-stream hazard (scripts/probes/two_stream_repro.py): an FFT-domain layer
    running while the split-fp16 convolutions (v_mfma_f32_16x16x32_f16) of a timing network run on a second HIP stream must
    return exactly what it returns alone.  With packed-FP32 vector instructions in the transform kernels it did not (8 of
    8 trials): those instructions return wrong values in lanes 48-63 while another dispatch's MFMA wave shares the SIMD;
    the kernels of the networks are built without them (build.py NO_PK)."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import fixture_waves as fw
    from amt_saga import _lib
    from amt_saga.device import stream_ptr
    from amt_saga.loop import TranscriptionLoop
    lib = _lib.load()
    B = 16
    rng = np.random.default_rng(3)
    cn = 32 if form == 'row' else 64
    k = (rng.standard_normal((4, 16, cn, cn)) * 0.05).astype(np.float32)
    s, t = np.ones(cn, np.float32), np.zeros(cn, np.float32)
    h = C.c_void_p()
    fp = lambda x: x.ctypes.data_as(C.c_void_p)
    create, destroy = (lib.amt_fftconv_create, lib.amt_fftconv_destroy) if form == 'row' else (lib.amt_fftpk_create, lib.amt_fftpk_destroy)
    _lib.check(create(C.byref(h), fp(k), fp(s), fp(t), fp(s), fp(t)))
    try:
        a = torch.rand((B, 20, 516, 32) if form == 'row' else (B, 10, 64, 64), device='cuda')
        o = torch.empty_like(a)
        need = lib.amt_fftconv_workspace_bytes(B, 20) if form == 'row' else lib.amt_fftpk_workspace_bytes(B)
        ws = torch.empty((need + 3) // 4, device='cuda')

        def run():
            if form == 'row':
                _lib.check(lib.amt_fftconv_run(h, a.data_ptr(), a.data_ptr(), B, 20, 516, o.data_ptr(), ws.data_ptr(), need, 1, stream_ptr()))
            else:
                _lib.check(lib.amt_fftpk_run(h, a.data_ptr(), a.data_ptr(), B, o.data_ptr(), ws.data_ptr(), need, 0, 1, stream_ptr()))
        c, p = fw.CASES['c3'], fw.params_for('c3')
        wave = torch.from_numpy(fw.pcm_to_wave(fw.render_pcm('c3', fw.note_lists('c3', B)))).cuda()
        lp = TranscriptionLoop(p, heads=('timing',), iters=1, groups=c['groups'], subtract=False).setup_device()
        batch = lp.prepare(wave)
        ct = batch.compress_bands(p.timing_bands, lp.refs['ref_mag'], p.timing_frames)
        net = lp.nets['timing_end']
        net.set_mode(2)                                        # split-fp16 direct convolutions only
        want_net = net.classify(ct).clone()
        run()
        torch.cuda.synchronize()
        want = o.clone()
        side, cur = torch.cuda.Stream(), torch.cuda.current_stream()
        for trial in range(6):
            o.zero_()
            torch.cuda.synchronize()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                y = net.classify(ct)
            run()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(o, want), (form, trial, float((o - want).abs().max()))
            assert torch.equal(y, want_net), (form, trial)
    finally:
        destroy(h)
