#!/usr/bin/env python3
"""Build libamt_saga_hip.so (gfx950) in-tree with hipcc.

    python amt-saga_amd/build.py [--force]

The shared library lands in amt-saga_amd/lib/ (git-ignored, travels with
gpurun snapshots).  No torch headers are involved: the C ABI (include/amt_saga.h)
takes plain device pointers.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libamt_saga_hip.so')
SOURCES = ['amt_stft.hip', 'amt_subtract.hip', 'amt_features.hip', 'amt_cqt.hip',
           'amt_rdcnn.hip', 'amt_loop.hip', 'amt_synth.hip', 'amt_train.hip', 'amt_probe.hip', 'amt_fftconv.hip', 'amt_fftpk.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math',
         '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
# per-file additions (appended, so they win).  The FFT butterflies and twiddle products of the
# STFT / iSTFT and the CQT accumulations are tolerance-checked floating point (1e-4 against the oracle): fused multiply-adds
# there cost nothing in parity and save ~a quarter of the vector instructions.  Everything that is
# compared bit for bit (subtract, the conv epilogues, the f32 MFMA chains) keeps contraction off.
# No packed-FP32 vector instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in any kernel of the networks: measured
# on MI355X (round 4, scripts/probes/two_stream_repro.py), they return wrong values in lanes 48-63 of a wave
# while a wave of ANOTHER dispatch issues back-to-back v_mfma_f32_16x16x32_f16 on the same SIMD -- i.e. whenever a
# transform kernel of one timing network shares CUs with the split-fp16 convolutions of the other (two streams).  With
# the instructions gone the same schedule is bit-identical to the serial one; the step time does not change (the
# transforms are not bound by vector issue).
NO_PK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
FILE_FLAGS = {'amt_stft.hip': ['-ffp-contract=fast'], 'amt_cqt.hip': ['-ffp-contract=fast'],
              'amt_fftconv.hip': NO_PK, 'amt_fftpk.hip': NO_PK, 'amt_rdcnn.hip': NO_PK}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
        [os.path.join(ROOT, 'include', 'amt_saga.h'), os.path.abspath(__file__)]      # (the flags live in this file)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + '.o')
        objs.append(o)
        if force or any(_newer(d, o) for d in [s] + deps[len(srcs):]):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
