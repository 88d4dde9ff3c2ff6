#!/usr/bin/env python3
"""Time of ONE packed-image FFT-domain 64 -> 64 (4 x 16) layer on 10 x 64 images (amt_fftpk_run) next to the direct
split-fp16 kernel's 0.91 ms per 1024 windows: GEMM alone (repeat), the chained layer (GEMM + inverse / register epilogue /
forward), the isolated layer.   python scripts/fftpk_bench.py [B=1024]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, W, Cn = 10, 64, 64
rng = np.random.default_rng(0)
k = (rng.standard_normal((4, 16, Cn, Cn)) * 0.04).astype(np.float32)
s1 = np.ones(Cn, np.float32); t1 = np.zeros(Cn, np.float32)
fp = lambda x: x.ctypes.data_as(C.c_void_p)
h = C.c_void_p()
_lib.check(lib.amt_fftpk_create(C.byref(h), fp(k), fp(s1), fp(t1), fp(s1), fp(t1)))
a = torch.rand((B, H, W, Cn), device='cuda')
sc = torch.rand((B, H, W, Cn), device='cuda')
out = torch.empty_like(a)
need = lib.amt_fftpk_workspace_bytes(B)
ws = torch.empty((need + 3) // 4, dtype=torch.float32, device='cuda')


def run(chain, rep, shortcut=True, n=5):
    args = (h, a.data_ptr(), sc.data_ptr() if shortcut else None, B, out.data_ptr(), ws.data_ptr(), need, chain, rep, None)
    _lib.check(lib.amt_fftpk_run(*args)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.check(lib.amt_fftpk_run(*args))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t1_, t5_ = run(0, 1), run(0, 5)
gemm = (t5_ - t1_) / 4
c0, c8 = run(0, 1, False), run(8, 1, False)
print('B %d: isolated layer (spatial in, shortcut, spatial out) %.3f ms; GEMM %.3f ms (%.2f TB/s of Xf + Yf); chained layer '
      '(GEMM + inverse / register epilogue / forward) %.3f ms' % (B, t1_, gemm, 2 * 577 * 128 * 4 * B / gemm / 1e9, (c8 - c0) / 8), flush=True)
