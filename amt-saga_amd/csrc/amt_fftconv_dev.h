// Device helpers shared by the FFT-domain convolution kernels (amt_fftconv.hip: row transforms of the 20 x 516 layers;
// amt_fftpk.hip: packed-image transforms of the 10 x 64 layers): the register-resident 24-point transform, the epilogue's
// sigmoid, scalar-base addressing, the split-fp16 operand exponent.
#pragma once
#include "amt_fft.h"

#define FC_LSCALE 2048.0f

typedef _Float16 fc_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 fc_h4 __attribute__((ext_vector_type(4)));
typedef float fc_f4 __attribute__((ext_vector_type(4)));

// e^{-2 pi i m / 24}, m = 0 .. 14
__device__ static const float fc_w24[15][2] = {
    {1.0f, 0.0f}, {0.96592582628906829f, -0.25881904510252076f}, {0.86602540378443865f, -0.5f},
    {0.70710678118654752f, -0.70710678118654752f}, {0.5f, -0.86602540378443865f},
    {0.25881904510252076f, -0.96592582628906829f}, {0.0f, -1.0f}, {-0.25881904510252076f, -0.96592582628906829f},
    {-0.5f, -0.86602540378443865f}, {-0.70710678118654752f, -0.70710678118654752f},
    {-0.86602540378443865f, -0.5f}, {-0.96592582628906829f, -0.25881904510252076f}, {-1.0f, 0.0f},
    {-0.96592582628906829f, 0.25881904510252076f}, {-0.86602540378443865f, 0.5f}};

// 24-point DFT in registers, natural order in and out: n = 3 n1 + n2, k = k1 + 8 k2; three 8-point transforms,
// twiddles W24^{n2 k1}, eight 3-point transforms.  INV: the conjugate transform (no 1 / 24).
template <bool INV>
__device__ __forceinline__ void fc_fft24(float2 (&x)[24]) {
    float2 y0[8], y1[8], y2[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) { y0[n1] = x[3 * n1]; y1[n1] = x[3 * n1 + 1]; y2[n1] = x[3 * n1 + 2]; }
    dft_r<8, INV>(y0); dft_r<8, INV>(y1); dft_r<8, INV>(y2);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
        const float2 w1 = make_float2(fc_w24[k1][0], INV ? -fc_w24[k1][1] : fc_w24[k1][1]);
        const float2 w2 = make_float2(fc_w24[2 * k1][0], INV ? -fc_w24[2 * k1][1] : fc_w24[2 * k1][1]);
        y1[k1] = cmul(y1[k1], w1);
        y2[k1] = cmul(y2[k1], w2);
    }
    const float s3 = 0.86602540378443865f;
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        const float2 a = y0[k1], b = y1[k1], c = y2[k1];
        const float2 t1 = cadd(b, c);
        const float2 t2 = make_float2(a.x - 0.5f * t1.x, a.y - 0.5f * t1.y);
        const float2 d = make_float2(s3 * (b.x - c.x), s3 * (b.y - c.y));
        // forward: X1 = t2 - i d, X2 = t2 + i d;  -i d = (d.y, -d.x)
        const float2 md = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        x[k1] = cadd(a, t1);
        x[k1 + 8] = cadd(t2, md);
        x[k1 + 16] = csub(t2, md);
    }
}

// 1 / (1 + e^-v): v_rcp_f32 + one Newton step instead of the IEEE division sequence (ten instructions, 48 times per
// thread and row: a sixth of the row kernel's vector instructions).  The step leaves the quotient within an ulp of the
// correctly rounded one; the argument is clamped so that 1 + e^-v stays finite (the step would turn 0 x inf into NaN).
// The epilogue's form: sigmoid(x / N * s1 + t1) = 1 / (1 + 2^(x k1 + k0)) with k1 = -s1 log2(e) / N and k0 = -t1 log2(e)
// formed once per thread and channel: one fma, a clamp, v_exp_f32, an add, v_rcp_f32 and the Newton step -- seven instructions
// per value where the unfolded form (two scalings, the BN multiply and add, the clamp, exp's own multiply) took ten.
__device__ __forceinline__ float fc_sigmoid_affine(float x, float k1, float k0) {
    const float d = 1.0f + __builtin_amdgcn_exp2f(fminf(fmaf(x, k1, k0), 125.0f));
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
#define FC_LOG2E 1.4426950408889634f
__device__ __forceinline__ float fc_sigmoid(float v) {
    const float d = 1.0f + __expf(-fmaxf(v, -87.0f));
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// uniform base + unsigned 32-bit byte offset: the access takes its base from scalar registers and ONE address VGPR (as
// 64-bit per-lane pointers the 72 addresses of a row thread alone overflowed the register file)
template <typename T> __device__ __forceinline__ const T &fc_at(const void *base, unsigned int off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const unsigned char *>(base) + off);
}
template <typename T> __device__ __forceinline__ T &fc_at(void *base, unsigned int off) {
    return *reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(base) + off);
}

__device__ __forceinline__ int fc_scale_exp(float amax) {
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &e);
    else return 0;
    return min(max(13 - e, -90), 90);
}


// a * conj(b)
__device__ __forceinline__ float2 fc_cmulc(float2 a, float2 b) {
    const amt_v2 t = amt_v2{a.y, a.y} * amt_v2{b.y, b.x};
    return f2(amt_v2{a.x, a.x} * amt_v2{b.x, -b.y} + t);
}

