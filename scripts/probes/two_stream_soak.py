#!/usr/bin/env python3
"""Soak of the opt-in two-stream mode at the metric size: the C3 step of bench.py's synthetic batch
(1024 windows of 516 frames by default), `reps` runs with timing_end on a second stream, every run compared bit for bit
(events AND the timing heads' floats) with the one-stream run.

    python scripts/probes/two_stream_soak.py [windows=1024] [reps=30]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import torch                                                 # noqa: E402
from amt_saga import synth                                   # noqa: E402
from amt_saga.hyperparams import Hyperparams                 # noqa: E402
from amt_saga.loop import TranscriptionLoop                  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
p = Hyperparams(N=2048)
lp = TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=1).setup_device()
L = p.H * (p.timing_frames - 1)
wave = synth.make_windows(B, L, seed=11, notes_per_window=(1, 4), device='cuda')[0]


def run(streams):
    lp.timing_streams = streams
    lp.trace = []
    ev, _ = lp.run(wave)
    torch.cuda.synchronize()
    tr, lp.trace = lp.trace, None
    return ev.clone(), [{k: v.clone() for k, v in t.items()} for t in tr]


ev1, tr1 = run(1)
ev1b, tr1b = run(1)
assert torch.equal(ev1, ev1b), 'the one-stream run is not reproducible'
bad = 0
for r in range(reps):
    ev2, tr2 = run(2)
    same = torch.equal(ev1, ev2) and all(torch.equal(a[k], b[k]) for a, b in zip(tr1, tr2) for k in a)
    bad += not same
    if not same:
        d = max(float((a[k] - b[k]).abs().max()) for a, b in zip(tr1, tr2) for k in a)
        print('run', r, 'differs: events equal', torch.equal(ev1, ev2), 'worst float distance', d, flush=True)
print('%d windows of %d frames: %d of %d two-stream runs differ from the one-stream run (events and head floats, bit for bit)'
      % (B, p.timing_frames, bad, reps), flush=True)
sys.exit(1 if bad else 0)
