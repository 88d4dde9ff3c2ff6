"""One GPU, world size 1, backend "nccl" (= RCCL on ROCm): the code path the 8-GPU scaling run takes --
init_process_group('nccl', device_id=...), the all-gather of DEVICE event tensors, the MAX / SUM reductions, the
barrier and the shutdown -- executed for real, in a fresh child process (the communicator must be created before
anything else touches the GPU in that process).  The gathered events must equal the plain single-process result.
The reference's parallelism is one worker per shard (training.py:623-634); here a shard is a rank's windows."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'amt-saga_amd')]
import numpy as np, torch
import torch.distributed as tdist
from amt_saga import dist as adist
rank, world, local = adist.init(force=True)
assert (rank, world) == (0, 1) and tdist.is_initialized() and tdist.get_backend() == 'nccl'
from amt_saga import synth
from amt_saga.hyperparams import Hyperparams
from amt_saga.loop import TranscriptionLoop
p = Hyperparams(N=2048, window_size_note_time=1)
B, iters = 6, 2
lo, hi = adist.shard_range(B, rank, world)
lp = TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=iters).setup_device()
L = p.H * (p.timing_frames - 1)
wave, _ = synth.make_windows(B, L, seed=5, notes_per_window=(1, 3), device='cuda')
events, b = lp.run(wave[lo:hi], window0=lo)
assert events.is_cuda
adist.barrier()
out = adist.gather_events(events, n_total=B * iters)          # device tensors through ncclAllGather
t = adist.max_over_ranks(3.25)
s = adist.sum_over_ranks(hi - lo)
# a larger device all-gather and an all-reduce on the same communicator (what a bigger job would move)
x = torch.arange(1 << 20, device='cuda', dtype=torch.float32)
parts = [torch.empty_like(x)]
tdist.all_gather(parts, x)
y = x.clone(); tdist.all_reduce(y)
torch.cuda.synchronize()
ok = bool(torch.equal(parts[0], x)) and bool(torch.equal(y, x))
adist.barrier()
adist.shutdown()
assert not tdist.is_initialized()
np.save(%(out)r, out.numpy())
np.save(%(out)r + '.plain.npy', events.reshape(-1, 7).cpu().numpy())
print(json.dumps(dict(max=t, sum=s, big_ok=ok, backend='nccl')))
'''


def test_rccl_world1_all_gather_of_device_events(tmp_path):
    out = str(tmp_path / 'ev.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, out=out))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('AMT_DIST_BACKEND', None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info == dict(max=3.25, sum=6.0, big_ok=True, backend='nccl')
    got, plain = np.load(out), np.load(out + '.plain.npy')
    key = plain[:, 0].astype(np.int64) * (1 << 20) + plain[:, 1]
    assert np.array_equal(got, plain[np.argsort(key, kind='stable')])
    assert got.shape == (12, 7) and np.array_equal(np.unique(got[:, 0]), np.arange(6))


WORKER2 = r'''
import os, sys, json
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'amt-saga_amd')]
import numpy as np, torch
import torch.distributed as tdist
from amt_saga import dist as adist
rank, world, local = adist.init()                              # before anything else touches the GPU in this process
assert world == 2 and tdist.get_backend() == 'nccl' and adist.world_size_seen() == 2
from amt_saga import synth
from amt_saga.hyperparams import Hyperparams
from amt_saga.loop import TranscriptionLoop
p = Hyperparams(N=2048, window_size_note_time=1)
B, iters = 12, 2
lo, hi = (0, 7) if rank == 0 else (7, 12)                      # ragged shards 7 + 5: the padded all-gather path
lp = TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=iters).setup_device()
L = p.H * (p.timing_frames - 1)
wave, _ = synth.make_windows(B, L, seed=5, notes_per_window=(1, 3), device='cuda')   # every rank renders all, runs its shard
events, b = lp.run(wave[lo:hi].contiguous(), window0=lo)
adist.barrier()
out = adist.gather_events(events, n_total=B * iters)
t = adist.max_over_ranks(float(rank + 1))
adist.barrier()
if rank == 0:
    np.save(%(out)r, out.numpy())
    print(json.dumps(dict(max=t, world=adist.world_size_seen())))
adist.shutdown()
'''


def test_rccl_world2_ragged_shards_equal_single_process(tmp_path):
    """Two ranks on two GPUs over RCCL (fresh child processes, created before any GPU call), ragged shards 7 + 5: the
    gathered events must equal the single-process run of the same 12 windows.  Skips on a one-GPU box -- the driver's
    multi-GPU node is where it runs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL world size 2)')
    from amt_saga import synth
    from amt_saga.hyperparams import Hyperparams
    from amt_saga.loop import TranscriptionLoop
    out = str(tmp_path / 'ev2.npy')
    script = tmp_path / 'worker2.py'
    script.write_text(WORKER2 % dict(root=ROOT, out=out))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('AMT_DIST_BACKEND', None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p_.communicate(timeout=900) for p_ in procs]
    assert all(p_.returncode == 0 for p_ in procs), [o[1][-3000:] for o in outs]
    info = json.loads(outs[0][0].strip().splitlines()[-1])
    assert info == dict(max=2.0, world=2)
    p = Hyperparams(N=2048, window_size_note_time=1)
    lp = TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=2).setup_device()
    wave, _ = synth.make_windows(12, p.H * (p.timing_frames - 1), seed=5, notes_per_window=(1, 3), device='cuda')
    plain = lp.run(wave)[0].reshape(-1, 7).cpu().numpy()
    key = plain[:, 0].astype(np.int64) * (1 << 20) + plain[:, 1]
    assert np.array_equal(np.load(out), plain[np.argsort(key, kind='stable')])
