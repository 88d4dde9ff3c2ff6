"""Time of the song-level CQT normalisers (amt_cqt_window_max) for one batch of full-size windows.
python scripts/cqt_max_bench.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'amt-saga_amd'))
import torch
from amt_saga import synth
from amt_saga.audio import cqt_table, cqt_window_max, midi_to_hz
from amt_saga.hyperparams import Hyperparams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = Hyperparams(N=2048)
L = p.H * (p.timing_frames - 1)
wave = synth.make_windows(B, L, 3, (2, 4), (0,), p.sr)[0]
f_lo = float(midi_to_hz(p.pitch_low))
span = p.pitch_high - p.pitch_low
for name, bpt in (('ref_C_1', 1), ('ref_C_inst', p.instrument_bins_per_tone), ('ref_C_foc', 4 * p.instrument_bins_per_tone)):
    tab = cqt_table(p.sr, f_lo, span * bpt, 12 * bpt, 'cuda')
    for form in ('valu', 'mfma'):
        cqt_window_max(wave, tab, p.H, form=form)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = cqt_window_max(wave, tab, p.H, form=form)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        sb = B * span * bpt * L
        print('%-10s %5d bins  %-4s form %7.1f ms / %d windows   %.2f T sample-bins/s   max %.4f' %
              (name, span * bpt, form, ms, B, sb / ms / 1e9, float(out.max())), flush=True)

# a whole song as one signal (transcribe.py / TranscriptionLoop.song_levels): 3 minutes
Ls = 180 * p.sr
song = (torch.rand(1, Ls, device='cuda') - 0.5) * 0.2
for name, bpt in (('ref_C_1', 1), ('ref_C_inst', p.instrument_bins_per_tone), ('ref_C_foc', 4 * p.instrument_bins_per_tone)):
    tab = cqt_table(p.sr, f_lo, span * bpt, 12 * bpt, 'cuda')
    cqt_window_max(song, tab, p.H)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = cqt_window_max(song, tab, p.H)
    torch.cuda.synchronize()
    print('song 180 s  %-10s %5d bins  %7.1f ms   max %.4f' % (name, span * bpt, (time.perf_counter() - t0) * 1e3, float(out[0])), flush=True)
