import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga.audio import cqt_slices, cqt_table, midi_to_hz
from amt_saga.hyperparams import Hyperparams
from amt_saga.device import to_dev
from oracle import cqt as ocqt
p = Hyperparams(N=2048)
B = 4
L = p.H * (p.timing_frames - 1)
torch.manual_seed(0)
w = torch.randn(B, L, device='cuda') * 0.1
src = to_dev(np.tile(np.arange(100, 108, dtype=np.int32)[None], (B, 1)), torch.int32)
f_lo = float(midi_to_hz(p.pitch_low))
tab = cqt_table(p.sr, f_lo, p.pitch_bands, 12 * p.pitch_bins_per_tone, 'cuda')
o = cqt_slices(w, src, tab, p.pitch_bands, p.H).cpu().numpy()
np.save(sys.argv[1], o)
t = ocqt.cqt_table(p.sr, f_lo, p.pitch_bands, 12 * p.pitch_bins_per_tone)
ref = ocqt.cqt_frames(w[0].cpu().numpy(), np.arange(100, 108), t[0], t[1], p.H)
print('vs oracle: max abs err / max', np.abs(o[0] - ref).max() / ref.max(), 'sum', o.sum(), 'ref sum(win0)', ref.sum(), o[0].sum())
