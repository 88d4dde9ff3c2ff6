// Shared device/host helpers for the gfx950 kernels of the AMT-SAGA hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "amt_saga.h"

#define AMT_WAVE 64

extern thread_local char amt_hip_err_buf[256];

#define AMT_HIP_CHECK(call)                                                        \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            snprintf(amt_hip_err_buf, sizeof(amt_hip_err_buf), "%s:%d %s -> %s",   \
                     __FILE__, __LINE__, #call, hipGetErrorString(e_));            \
            return AMT_E_HIP;                                                      \
        }                                                                          \
    } while (0)

#define AMT_LAUNCH_CHECK() AMT_HIP_CHECK(hipGetLastError())

// ---- complex helpers ---------------------------------------------------------
// Written on clang's native 2-vectors so that the backend selects the packed FP32 instructions of
// CDNA3/4 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: both halves of a complex number per lane and
// instruction).  The FFT kernels are bound by vector-instruction issue, not by LDS or HBM (a 2048-point
// transform costs ~2600 issue cycles per SIMD in scalar form), so halving the instruction count of the
// butterflies is what moves them.
typedef float amt_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ amt_v2 v2(float2 a) { return amt_v2{a.x, a.y}; }
__device__ __forceinline__ float2 f2(amt_v2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return f2(v2(a) + v2(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return f2(v2(a) - v2(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    // (ax bx - ay by, ax by + ay bx) = ax * (bx, by) + ay * (-by, bx)
    const amt_v2 t = amt_v2{a.y, a.y} * amt_v2{-b.y, b.x};
    return f2(amt_v2{a.x, a.x} * v2(b) + t);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return f2(v2(a) * amt_v2{s, s}); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// multiply by -i : (x + iy)(-i) = y - ix
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }
// multiply by +i
__device__ __forceinline__ float2 cmul_pi(float2 a) { return make_float2(-a.y, a.x); }

// ---- reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Order-preserving float <-> uint map so that unsigned atomicMax implements a
// float max for any sign (relu=False leaves negatives in the residual).
__device__ __forceinline__ unsigned int float_to_ordered(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
#define AMT_ORDERED_NEG_INF 0x007fffffu   /* float_to_ordered(-inf) */

// block max (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float block_max(float v, float *lds_scratch /* >= 16 floats */) {
    v = wave_max(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds_scratch[wid] = v;
    __syncthreads();
    if (wid == 0) {
        const int nw = blockDim.x >> 6;
        float r = lane < nw ? lds_scratch[lane] : -INFINITY;
        r = wave_max(r);
        v = r;
    }
    return v;
}

static inline int amt_is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
