import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import heads
from amt_saga.hyperparams import Hyperparams
p = Hyperparams(N=2048)
rng = np.random.default_rng(0)
import inspect
print(inspect.signature(heads.InstrumentClassifier.__init__))
for variant in ('instrument', 'instrument_focused', 'instrument_dual'):
    try:
        h = heads.InstrumentClassifier(p, variant)
    except Exception as e:
        print(variant, 'ctor:', e); continue
    shapes = h.cfg['input_shapes']
    B = 8
    xs = [torch.from_numpy((rng.random((B,) + tuple(s[:2])) ** 2).astype(np.float32)).cuda() for s in shapes]
    y = rng.integers(0, h.cfg['output_classes'], B)
    x = xs if len(xs) > 1 else xs[0]
    for i in range(3):
        h.train(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        h.train(x, y)
    torch.cuda.synchronize()
    print(variant, shapes, 'loss', [round(m[0], 4) for m in h.metrics_train[-3:]], 'ms/step %.2f' % ((time.perf_counter() - t0) / 5 * 1e3))
