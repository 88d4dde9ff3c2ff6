"""compress_bands kernel time (1024 windows, 516 frames, 20 bands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch
from amt_saga.audio import AudioBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = torch.randn(B, 512 * 515, device='cuda') * 0.1
b = AudioBatch(w, 2048, 512).stft(False)
for _ in range(2): b.compress_bands(20, b.ref_max, 516)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): o = b.compress_bands(20, b.ref_max, 516)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('compress_bands B=%d %.3f ms  %.2f TB/s' % (B, ms, B * 516 * 1028 * 4 / ms / 1e9), 'checksum', float(o.double().sum()))
