"""CPU, world_size 2 over gloo: the N>1 path -- contiguous window shards, ragged
event all-gather, order/determinism vs the single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import dist as adist
rank, world, local = adist.init(backend='gloo')
assert world == %(world)d
B, iters = %(B)d, 3
lo, hi = adist.shard_range(B, rank, world)
# deterministic fake events for the windows of this shard
w = np.arange(lo, hi)
ev = np.zeros((iters, hi - lo, 7), np.int32)
for it in range(iters):
    ev[it, :, 0] = w; ev[it, :, 1] = it; ev[it, :, 2] = 21 + (w * 7 + it) %% 88
    ev[it, :, 3] = (w * 3) %% 112; ev[it, :, 4] = 1 + (w + it) %% 127; ev[it, :, 5] = w %% 516; ev[it, :, 6] = (w + 9) %% 517
out = adist.gather_events(torch.from_numpy(ev).reshape(-1, 7), n_total=B * iters)
t = adist.max_over_ranks(1.0 + rank)
s = adist.sum_over_ranks(hi - lo)
adist.barrier()
if rank == 0:
    np.save(%(out)r, out.numpy())
    print('MAX', t, 'SUM', s)
adist.shutdown()
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _expected(B, iters=3):
    w = np.repeat(np.arange(B), iters)
    it = np.tile(np.arange(iters), B)
    ev = np.stack([w, it, 21 + (w * 7 + it) % 88, (w * 3) % 112, 1 + (w + it) % 127, w % 516,
                   (w + 9) % 517], axis=1).astype(np.int32)
    return ev


@pytest.mark.parametrize('world,B', [(2, 10), (2, 7), (3, 4)])
def test_gather_events_gloo(tmp_path, world, B):
    out = str(tmp_path / 'ev.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, world=world, B=B, out=out))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = np.load(out)
    assert np.array_equal(got, _expected(B))            # sorted by (window, iter), same as 1 process
    assert 'MAX %s SUM %s' % (float(world), float(B)) in outs[0]


def test_shard_range():
    from amt_saga.dist import shard_range
    for B, G in ((1024, 4), (16384, 8), (10, 3), (3, 8), (0, 2)):
        spans = [shard_range(B, r, G) for r in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) == -(-B // G)
    assert shard_range(16384, 3, 8) == (6144, 8192)


def test_single_process_gather_is_identity():
    import torch
    from amt_saga.dist import gather_events
    ev = torch.from_numpy(_expected(5)[::-1].copy())
    assert np.array_equal(gather_events(ev).numpy(), _expected(5))
