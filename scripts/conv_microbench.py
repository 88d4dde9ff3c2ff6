#!/usr/bin/env python3
"""Per-layer timing of one RDCNN head in both convolution arithmetics (HIP events
from amt_rdcnn_profile).  python scripts/conv_microbench.py [timing|pitch|...] [B] [modes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch
from amt_saga import heads
from amt_saga.hyperparams import Hyperparams

name = sys.argv[1] if len(sys.argv) > 1 else 'timing'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else '0,1').split(',')]
cal = (sys.argv[4] if len(sys.argv) > 4 else 'cal') == 'cal'      # calibrated (input-sensitive) or raw random BN statistics
p = Hyperparams(N=2048)
h = {'timing': heads.timming_classifier, 'pitch': heads.pitch_classifier,
     'velocity': heads.VelocityClassifier}[name](p, calibrated=cal)
print('head', name, 'B', B, 'calibrated', h.calibrated)
H, W, _ = h.cfg['input_shapes'][0]
x = torch.rand(B, H, W, device='cuda') ** 2
for mode in modes:
    h.set_mode(mode)
    h.predict_device([x]); torch.cuda.synchronize()
    h.profile(True); h.profile_read(reset=True)
    for _ in range(2):
        h.predict_device([x])
    torch.cuda.synchronize()
    rows = h.profile_read(reset=True)
    h.profile(False)
    tot = sum(r['ms'] for r in rows)
    fl = sum(r['flops_per_window'] * r['windows'] for r in rows)
    print('mode', mode, 'conv total ms/forward %.2f  %.1f TFLOP/s' % (tot / 2, fl / tot / 1e9))
    for r in rows:
        if r['layer'] in (1, 2, 3, 12, 13, 14, 24, 25, 26, 33):
            print('  L%-2d %dx%d %3d->%3d  %8.3f ms  %6.1f TF/s' % (
                r['layer'], r['H'], r['W'], r['cin'], r['cout'], r['ms'] / 2,
                r['flops_per_window'] * r['windows'] / r['ms'] / 1e9))
