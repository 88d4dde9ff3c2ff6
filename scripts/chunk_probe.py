#!/usr/bin/env python3
"""Per-layer-class time of the timing head (conv mode 3) for one setting of AMT_RD_CHUNK (read by the library at first use):
python scripts/chunk_probe.py [B=1024].  Prints ms per 1024 windows for the FFT-domain layers, the 10 x 64 and the 5 x 8 classes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch
from amt_saga import heads
from amt_saga.hyperparams import Hyperparams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = Hyperparams(N=2048)
h = heads.timming_classifier(p, calibrated=True)
H, W, _ = h.cfg['input_shapes'][0]
x = torch.rand(B, H, W, device='cuda') ** 2
h.set_mode(3)
h.predict_device([x]); torch.cuda.synchronize()
h.profile(True); h.profile_read(reset=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(3):
    h.predict_device([x])
ev1.record()
torch.cuda.synchronize()
rows = h.profile_read(reset=True)
h.profile(False)
cls = {}
for r in rows:
    k = '%dx%d' % (r['H'], r['W']) + (' L1' if r['layer'] == 1 else '')
    cls[k] = cls.get(k, 0.0) + r['ms'] / 3
print('AMT_RD_CHUNK', os.environ.get('AMT_RD_CHUNK', '-'), 'B', B, 'forward %.2f ms;' % (ev0.elapsed_time(ev1) / 3),
      '; '.join('%s %.2f' % kv for kv in cls.items()), flush=True)
