"""CPU: the C-ABI library loads and exports every symbol include/amt_saga.h
declares (no compute calls without a GPU); the product path fails loudly
without a GPU / without the library; the oracle is never imported by it."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'amt_saga.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(amt_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    from amt_saga import _lib
    names = _declared()
    assert len(names) >= 20
    assert sorted(_lib.PROTOTYPES) == names, set(names) ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    lib2 = _lib.load()
    assert lib2.amt_version() == 1
    assert lib2.amt_strerror(-2) == b'Invalid Input shape'
    assert lib2.amt_strerror(-6) == b'Requested attribute does not exist'


def test_host_side_argument_checks_without_gpu():
    """Pure host validation paths of the ABI (they return before touching HIP)."""
    from amt_saga import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.amt_stft_plan_create(ctypes.byref(h), 300, 75, 1) == _lib.AMT_E_INVALID
    assert lib.amt_stft_plan_create(ctypes.byref(h), 8192, 2048, 1) == _lib.AMT_E_INVALID
    assert lib.amt_stft_frames(None, 100) == _lib.AMT_E_INVALID
    assert lib.amt_subtract(None, None) == _lib.AMT_E_INVALID
    assert lib.amt_cqt_slices(None, None) == _lib.AMT_E_INVALID
    d = _lib.RdcnnDesc()
    assert lib.amt_rdcnn_param_count(ctypes.byref(d)) == 0          # n_towers = 0 -> invalid
    with pytest.raises(ValueError):
        _lib.check(_lib.AMT_E_SHAPE)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.AMT_E_HIP)


def test_param_counts_match_survey():
    """Python topology walk == C++ topology walk == SURVEY 3.3 table."""
    from amt_saga import _lib, heads
    from amt_saga.hyperparams import Hyperparams
    lib = _lib.load()
    p = Hyperparams(N=4096)
    expect = {'pitch': (2560, 2.38e6), 'instrument': (1280, 2.03e6), 'velocity': (2304, 0.94e6),
              'timing': (2560, 13.4e6)}
    made = {'pitch': heads.pitch_classifier(p), 'instrument': heads.InstrumentClassifier(p, 'instrument'),
            'velocity': heads.VelocityClassifier(p), 'timing': heads.timming_classifier(p)}
    for k, h in made.items():
        flat, n = expect[k]
        assert h.flat == flat
        assert abs(h._blob.size - n) / n < 0.01
        d = h._desc()
        assert lib.amt_rdcnn_param_count(ctypes.byref(d)) == h._blob.size
    dual = heads.InstrumentClassifier(p, 'instrument_dual')
    assert dual.flat == 2560 and len(dual.cfg['input_shapes']) == 2
    p2 = Hyperparams(N=2048)
    assert heads.timming_classifier(p2).flat == 5120
    with pytest.raises(ValueError):
        heads.InstrumentClassifier(p, 'Invalid')


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from amt_saga import audio, heads
    from amt_saga.hyperparams import Hyperparams
    with pytest.raises(RuntimeError):
        audio.AudioBatch(np.zeros((1, 4096), np.float32), 512)
    with pytest.raises(RuntimeError):
        audio.audio_complete(np.zeros(4096, np.float32), 512).mag
    with pytest.raises(RuntimeError):
        heads.VelocityClassifier(Hyperparams()).classify(np.zeros((36, 8), np.float32))


def test_product_never_imports_oracle():
    code = ("import sys; sys.path[:0]=[%r, %r]; import amt_saga, amt_saga.audio, amt_saga.rdcnn, "
            "amt_saga.heads, amt_saga.loop, amt_saga.dist, amt_saga.synth; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
            % (ROOT, os.path.join(ROOT, 'amt-saga_amd')))
    subprocess.check_call([sys.executable, '-c', code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'amt-saga_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from amt_saga import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_split_bf16_kernels_are_spill_free():
    """The split-bf16 / split-fp16 conv kernels prefetch weights with inline-asm loads whose
    destination registers must never be spilled while in flight (a spill would save
    stale data): require zero VGPR spills / scratch for every instantiation."""
    import shutil
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'amt-saga_amd', 'csrc', 'amt_rdcnn.hip')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math',
           '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'),
           '-I' + os.path.join(ROOT, 'amt-saga_amd', 'csrc'), '-c', src, '-o', os.devnull,
           '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = out.split('Function Name: ')[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not any(k in name for k in ('conv_bf16x6_kernel', 'conv_f16x3s_kernel')):
            continue
        seen += 1
        spill = int(re.search(r'VGPRs Spill: (\d+)', b).group(1))
        scratch = int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1))
        vgpr = int(re.search(r'\bVGPRs: (\d+)', b).group(1))
        assert spill == 0 and scratch == 0, (name, spill, scratch)
        # 4 waves per SIMD = two 512-thread workgroups per CU; the diagnostic four-subtile form (template
        # argument MS = 4, 256-thread workgroups) runs two waves per SIMD by design
        # (mangled template tail: ... MASKED, MS, TRAIN -> 'Lb?ELi<MS>ELb?EEv')
        ms4 = 'conv_f16x3s_kernel' in name and re.search(r'ELi4ELb[01]EEv', name) is not None
        assert vgpr <= (256 if ms4 else 128), (name, vgpr)
    assert seen >= 24 + 18 + 18                     # + the trainer's forms (TRAIN = true) of the split-fp16 kernel
