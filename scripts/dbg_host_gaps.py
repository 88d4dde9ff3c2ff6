"""Host-side time of the phases of one C3 step (no GPU sync inside the phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch
from amt_saga import synth, dist as adist
from amt_saga.hyperparams import Hyperparams
from amt_saga.loop import TranscriptionLoop
from amt_saga.device import empty
p = Hyperparams(N=2048)
loop = TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=1).setup_device()
B = 1024
L = p.H * (p.timing_frames - 1)
wave, _ = synth.make_windows(B, L, seed=3000, notes_per_window=(3, 3), device='cuda')
loop.prepare(wave); refs = loop.refs
for s in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = loop.prepare(wave, refs)
    t1 = time.perf_counter()
    events = empty((1, B, 7), torch.int32)
    loop.iterate(b, 0, events, 0)
    t2 = time.perf_counter()
    ev = adist.gather_events(events.reshape(-1, 7), n_total=B)
    t3 = time.perf_counter()
    del b, events
    t4 = time.perf_counter()
    print('step %d host ms: prepare %.2f  iterate(launch) %.2f  gather(sync) %.2f  free %.2f' % (
        s, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
