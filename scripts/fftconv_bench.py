#!/usr/bin/env python3
"""Time of ONE FFT-domain 32 -> 32 (4 x 16) layer on 20 x 516 images (amt_fftconv_run: forward FFT, GEMM, inverse +
epilogue, spatial in / spatial out) next to the direct split-fp16 kernel's 3.3 ms per 1024 windows.
python scripts/fftconv_bench.py [B=1024]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, W = 20, 516
rng = np.random.default_rng(0)
k = (rng.standard_normal((4, 16, 32, 32)) * 0.05).astype(np.float32)
s1 = np.ones(32, np.float32); t1 = np.zeros(32, np.float32)
fp = lambda x: x.ctypes.data_as(C.c_void_p)
h = C.c_void_p()
_lib.check(lib.amt_fftconv_create(C.byref(h), fp(k), fp(s1), fp(t1), fp(s1), fp(t1)))
a = torch.rand((B, H, W, 32), device='cuda')
sc = torch.rand((B, H, W, 32), device='cuda')
out = torch.empty_like(a)
need = lib.amt_fftconv_workspace_bytes(B, H)
ws = torch.empty((need + 3) // 4, dtype=torch.float32, device='cuda')


def run(rep, n=5):
    _lib.check(lib.amt_fftconv_run(h, a.data_ptr(), sc.data_ptr(), B, H, W, out.data_ptr(), ws.data_ptr(), need, rep, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.check(lib.amt_fftconv_run(h, a.data_ptr(), sc.data_ptr(), B, H, W, out.data_ptr(), ws.data_ptr(), need, rep, None))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t1_, t5_ = run(1), run(5)
gemm = (t5_ - t1_) / 4
print('B %d: layer (isolated: spatial in, spatial out) %.2f ms; GEMM %.2f ms; forward FFT + inverse/epilogue %.2f ms' %
      (B, t1_, gemm, t1_ - gemm), flush=True)
fl = 2.0 * 289 * B * H * 256 * 64
print('GEMM: %.1f TFLOP/s f32-equivalent (x3 executed), %.2f TB/s of Xf + Yf' % (fl / gemm / 1e9, 2 * 289 * B * H * 64 * 4 / gemm / 1e9))
