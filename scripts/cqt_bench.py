"""CQT slice kernel time (1024 windows): pitch (174 bins), instrument (348 bins), velocity (36 bins)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga.audio import cqt_slices, cqt_table, midi_to_hz
from amt_saga.hyperparams import Hyperparams
from amt_saga.device import to_dev
p = Hyperparams(N=2048)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = p.H * (p.timing_frames - 1)
w = torch.randn(B, L, device='cuda') * 0.1
src = to_dev(np.tile(np.arange(100, 108, dtype=np.int32)[None], (B, 1)), torch.int32)
f_lo = float(midi_to_hz(p.pitch_low))
for name, bands, bpo in (('pitch', p.pitch_bands, 12 * p.pitch_bins_per_tone),
                         ('instrument', p.instrument_bands, 12 * p.instrument_bins_per_tone)):
    tab = cqt_table(p.sr, f_lo, bands, bpo, 'cuda')
    for _ in range(2): o = cqt_slices(w, src, tab, bands, p.H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): o = cqt_slices(w, src, tab, bands, p.H)
    e1.record(); torch.cuda.synchronize()
    print('cqt %-10s bins %3d  %.3f ms  checksum %.6e' % (name, bands, e0.elapsed_time(e1) / 3, float(o.double().sum())))
