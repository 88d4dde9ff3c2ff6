"""Timing breakdown of the MFMA CQT-max kernel (AMT_CQM_DEBUG knobs; results are garbage under them)."""
import os, sys, time, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'amt-saga_amd'))
    import torch
    from amt_saga import synth
    from amt_saga.audio import cqt_table, cqt_window_max, midi_to_hz
    from amt_saga.hyperparams import Hyperparams
    p = Hyperparams(N=2048)
    L = p.H * (p.timing_frames - 1)
    wave = synth.make_windows(1024, L, 3, (2, 4), (0,), p.sr)[0]
    tab = cqt_table(p.sr, float(midi_to_hz(p.pitch_low)), (p.pitch_high - p.pitch_low) * 16, 192, 'cuda')
    cqt_window_max(wave, tab, p.H, form='mfma'); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        cqt_window_max(wave, tab, p.H, form='mfma')
    torch.cuda.synchronize()
    print('AMT_CQM_DEBUG=%s: %.1f ms' % (os.environ.get('AMT_CQM_DEBUG', '0'), (time.perf_counter() - t0) / 3 * 1e3), flush=True)
else:
    for d in ('0', '1', '2', '3', '5', '6', '7'):
        subprocess.run([sys.executable, os.path.abspath(__file__), 'run'], env=dict(os.environ, AMT_CQM_DEBUG=d))
