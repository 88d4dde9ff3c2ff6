"""Torch plumbing: device memory, streams, pointers.  Not the product."""
import ctypes as C

import numpy as np
import torch


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('amt_saga: no GPU visible (torch.cuda.is_available() is False); '
                           'the HIP path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous tensor (or None)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), 'non-contiguous tensor at the C ABI'
    return C.c_void_p(t.data_ptr())


def to_dev(a, dtype=torch.float32, device=None):
    device = device or require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), device='cpu').to(device=device, dtype=dtype).contiguous()


def empty(shape, dtype=torch.float32, device=None):
    return torch.empty(shape, dtype=dtype, device=device or require_gpu())


def zeros(shape, dtype=torch.float32, device=None):
    return torch.zeros(shape, dtype=dtype, device=device or require_gpu())
