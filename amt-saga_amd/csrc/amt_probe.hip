// Measurement probes of the chip itself (no reference counterpart: bench.py's roofline denominators).
//
// amt_probe_mfma_f16: every wave issues v_mfma_f32_16x16x32_f16 back to back on register operands -- no LDS,
// no memory -- for a few tens of milliseconds; returns the sustained TFLOP/s.  The matrix pipe's power draw
// depends on the operand bits and the clock the power limit allows sets the rate (MI355X_MICROARCH.md "DVFS
// give-back"), so the probe runs on random non-zero operands by default; zeros give the cycle-limited figure.
// bench.py calls it in the same process as the timed steps, so "executed MFMA rate vs what this chip sustains"
// is a ratio of two numbers measured in one run on one device (round 2 carried a literal copied from a profile).
#include "amt_common.h"

typedef _Float16 pr_h8 __attribute__((ext_vector_type(8)));
typedef float pr_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe_fill_kernel(pr_h8 *a, int n, int random) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    pr_h8 v;
    unsigned s = 0x9E3779B9u * (unsigned)(t + 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f;       // [-0.5, 0.5)
        v[e] = random ? (_Float16)(u * 0.25f) : (_Float16)0.f;
    }
    a[t] = v;
}

__global__ __launch_bounds__(256) void probe_mfma_f16_kernel(const pr_h8 *__restrict__ a_in, float *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const pr_h8 a0 = a_in[t], b0 = a_in[t ^ 2], a1 = a_in[t ^ 1], b1 = a_in[t ^ 3];
    pr_f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c7, 0, 0, 0);
    }
    const pr_f4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[t] = s.x + s.y + s.z + s.w;
}

extern "C" int amt_probe_mfma_f16(int waves_per_simd, int iters, int random_operands, int launches,
                                  double *tflops_best, double *ms_best, void *stream) {
    if (!tflops_best || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || launches < 1) return AMT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    AMT_HIP_CHECK(hipGetDevice(&dev));
    AMT_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * waves_per_simd, n = blocks * 256;
    pr_h8 *a = nullptr;
    float *out = nullptr;
    AMT_HIP_CHECK(hipMalloc(&a, (size_t)n * sizeof(pr_h8)));
    if (hipMalloc(&out, (size_t)n * sizeof(float)) != hipSuccess) { (void)hipFree(a); return AMT_E_NOMEM; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe_fill_kernel<<<blocks, 256, 0, st>>>(a, n, random_operands);
    probe_mfma_f16_kernel<<<blocks, 256, 0, st>>>(a, out, iters / 16 + 1);          // warm the clocks
    double best = 0.0, best_ms = 0.0;
    int rc = AMT_OK;
    for (int r = 0; r < launches && rc == AMT_OK; ++r) {
        (void)hipEventRecord(e0, st);
        probe_mfma_f16_kernel<<<blocks, 256, 0, st>>>(a, out, iters);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = AMT_E_HIP; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * (double)iters * 8 * (2.0 * 16 * 16 * 32);
        const double tf = fl / (ms * 1e-3) / 1e12;
        if (tf > best) { best = tf; best_ms = ms; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(out);
    *tflops_best = best;
    if (ms_best) *ms_best = best_ms;
    return rc;
}
