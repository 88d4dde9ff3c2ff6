"""STFT kernel rate: algorithmic bytes (4L read, mag 4FT (+ phase 8FT) written per window) / time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch
from amt_saga.audio import AudioBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = 512 * 515
w = torch.randn(B, L, device='cuda') * 0.1
for ph in (True, False):
    b = AudioBatch(w, 2048, 512)
    for _ in range(2): b.stft(ph)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): b.stft(ph)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byt = B * (4 * L + (12 if ph else 4) * 1025 * 516)
    print('stft phase=%s B=%d  %.3f ms  %.2f TB/s algorithmic' % (ph, B, ms, byt / ms / 1e9))
