"""Is the STFT kernel's duration data dependent?  HIP-synth windows vs CPU-synth vs noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import synth
from amt_saga.audio import AudioBatch
B, L = 256, 512 * 515
def t(wave, tag):
    for _ in range(2):
        AudioBatch(wave, 2048, 512).stft(with_phase=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b = AudioBatch(wave, 2048, 512).stft(with_phase=True)
    e1.record(); torch.cuda.synchronize()
    print(tag, 'ms/stft', e0.elapsed_time(e1) / 5, 'finite', bool(torch.isfinite(wave).all()),
          'absmax', float(wave.abs().max()), 'tiny', int(((wave != 0) & (wave.abs() < 1.2e-38)).sum()),
          'small', int(((wave != 0) & (wave.abs() < 1e-30)).sum()), flush=True)
w_hip, notes = synth.make_windows(B, L, seed=3000, notes_per_window=(3, 3), device='cuda')
t(w_hip, 'hip-synth')
w_cpu = torch.stack([synth.render_window(ns, L) for ns in notes[:8]]).cuda()
print('hip vs cpu synth maxdiff', float((w_hip[:8] - w_cpu).abs().max()))
t(w_cpu.repeat(B // 8, 1).contiguous(), 'cpu-synth')
t(torch.randn(B, L, device='cuda') * 0.01, 'noise')
t(torch.zeros(B, L, device='cuda'), 'zeros')
