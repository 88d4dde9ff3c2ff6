#!/usr/bin/env python3
"""Training-step time of the RDCNN heads (amt_trainer_step: training-mode BN, backward, Adagrad).
python scripts/train_bench.py [head=timing|pitch|velocity] [batch=8] [steps=5]
Reports ms per step, windows/s and the f32-MFMA rate of the step (3 x the forward flops: forward, data
gradient, weight gradient).  The reference trains with batch 8 (util_train_test.py: batch_size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np
import torch
from amt_saga import heads
from amt_saga.hyperparams import Hyperparams

name = sys.argv[1] if len(sys.argv) > 1 else 'timing'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
p = Hyperparams(N=2048)
h = {'timing': heads.timming_classifier, 'pitch': heads.pitch_classifier,
     'velocity': heads.VelocityClassifier}[name](p)
H, W, _ = h.cfg['input_shapes'][0]
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.random((B, H, W)) ** 2).astype(np.float32)).cuda()
lo, hi = h.cfg['output_range'] if h.cfg['output_classes'] == 1 else (0, h.cfg['output_classes'])
y = rng.uniform(lo, hi, B) if h.cfg['output_classes'] == 1 else rng.integers(0, h.cfg['output_classes'], B)
h.train(x, y); h.train(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    h.train(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
h._ensure()
fl = 3.0 * h.flops_per_window * B
print('train %-8s batch %3d: %8.2f ms/step  %7.1f windows/s  %6.1f TFLOP/s (3 x forward flops; f32 MFMA peak 157)  loss %.4f' %
      (name, B, dt * 1e3, B / dt, fl / dt / 1e12, h.metrics_train[-1][0]), flush=True)
