"""Multi-GPU: one process per GPU, windows sharded, events all-gathered.

The reference's only parallelism is multiprocessing on one host (one song per
worker, training.py:623-634).  Here independent windows shard across the GPUs
of a node in contiguous blocks; weights, CQT tables and the guess
bank are replicated; nothing is exchanged inside the loop.  The one collective
is an all-gather of the fixed-size event records {window, iter, pitch, program,
velocity, onset_frame, end_frame} (7 x int32) over RCCL/xGMI (backend "nccl" on
ROCm) -- a few hundred KB, latency-bound.  On CPU (tests) the same code runs
over gloo.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)),
            int(os.environ.get('LOCAL_RANK', 0)))


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (no-op for a
    single process unless force=True or AMT_DIST_FORCE=1: then a world of ONE rank
    still creates the process group, so that the RCCL communicator, the device-tensor
    all-gather and the reductions below execute on a one-GPU box).
    Returns (rank, world, local_rank)."""
    rank, world, local = env_world()
    force = force or os.environ.get('AMT_DIST_FORCE') == '1'
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # AMT_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)
            backend = os.environ.get('AMT_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        if backend == 'nccl':
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    return rank, world, local


def world_size_seen():
    """The size of the process group as torch.distributed reports it after init (1 without a group): what bench.py
    prints as `rccl_world`, so that a scaling record shows how many ranks really ran."""
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def shard_range(n_windows, rank, world):
    """Contiguous block of ceil(B/G) windows per rank; the last
    ranks may get fewer (or none)."""
    per = -(-n_windows // world)
    lo = min(rank * per, n_windows)
    return lo, min(lo + per, n_windows)


def gather_events(events, n_total=None):
    """All-gather the per-rank event records.  events: int32 [n_local, 7] (any
    device).  Ragged shards are padded to the largest shard for the collective
    and trimmed afterwards.  Returns int32 [sum n_local, 7] sorted by
    (window, iter) -- identical for every world size (determinism check)."""
    ev = events.reshape(-1, events.shape[-1]).contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        out = ev
    else:
        world = dist.get_world_size()
        if dist.get_backend() != 'nccl':
            ev = ev.cpu()                      # gloo collectives run on host tensors
        n = torch.tensor([ev.shape[0]], dtype=torch.int64, device=ev.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
        mx = max(counts)
        pad = torch.full((mx, ev.shape[1]), -1, dtype=ev.dtype, device=ev.device)
        pad[:ev.shape[0]] = ev
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    out = out.cpu()
    key = out[:, 0].to(torch.int64) * (1 << 20) + out[:, 1].to(torch.int64)
    out = out[torch.argsort(key, stable=True)]
    if n_total is not None and out.shape[0] != n_total:
        raise RuntimeError('gathered %d events, expected %d' % (out.shape[0], n_total))
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x, device=None):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if (
            dist.get_backend() == 'nccl') else torch.device('cpu')
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if (
            dist.get_backend() == 'nccl') else torch.device('cpu')
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
