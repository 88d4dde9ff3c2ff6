// Packed-FP32 vector instructions beside v_mfma_f32_16x16x32_f16 on MI355X (gfx950): a self-checking probe.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/pk_mfma_hazard.hip -o /tmp/pk_mfma_hazard && /tmp/pk_mfma_hazard
// Round 4 found the FFT-domain transform kernels returning wrong values whenever they shared CUs with the split-fp16
// convolutions of ANOTHER stream.  This probe removes everything of the product: a "victim" kernel issues one
// packed instruction (v_pk_mul_f32 plain / with the op_sel + neg modifiers of a complex multiply / v_pk_fma_f32 /
// v_pk_add_f32) and the two scalar instructions that compute the same thing (v_mul_f32 ...) on the same registers in a
// loop and counts, per lane, how often the two disagree bit for bit; an "aggressor" kernel issues back-to-back
// v_mfma_f32_16x16x32_f16 on register operands (no LDS, no memory).  Legs: the victim alone; victim and aggressor on two
// streams; both roles in ONE dispatch (odd workgroups multiply matrices, even ones are victims); the scalar control.
//
// RESULT (round 4, profiles/r04/pk_mfma_hazard_probe.txt): NO mismatch in any leg -- 1e10 checks each, also for the
// dependent mul / mul / add chain of the transforms' complex multiply with its second operand freshly read from LDS.
// An isolated packed instruction beside MFMA waves is therefore fine; what is established at the product level
// (scripts/probes/two_stream_repro.py, tests/test_gpu_fftconv.py) is narrower: the transform kernels compiled WITH
// packed-FP32 instructions are wrong in 8 of 8 concurrent trials (lanes 48-63 of one LDS write of twiddle products),
// the same sources compiled WITHOUT them (-target-feature -packed-fp32-ops) in 0 of 16 and in none of the tests, and a
// build that only un-packs the twiddle products (SLP vectoriser off, 378 of 467 packed instructions left in the
// butterflies) in 0 of 8.  The exact trigger inside those kernels is not isolated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ unsigned victim_loop(int iters) {
    __shared__ f2 lds[256];
    const int t = threadIdx.x;
    f2 a = {1.0f + 0.001f * t, 0.5f + 0.002f * t}, b = {0.75f + 0.003f * t, 1.25f - 0.001f * t}, c = {0.1f * (t & 7), -0.05f * (t & 3)};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 r;
        float rx, ry;
        if (KIND == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rx) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.y));
        } else if (KIND == 1) {            // the first product of a complex multiply: lo = a.y * -b.y, hi = a.y * b.x
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            asm volatile("v_mul_f32 %0, %1, -%2" : "=v"(rx) : "v"(a.y), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.x));
        } else if (KIND == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(rx) : "v"(a.x), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ry) : "v"(a.y), "v"(b.y), "v"(c.y));
        } else if (KIND == 3) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(rx) : "v"(a.x), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.y));
        } else if (KIND == 5 || KIND == 6) {   // the complex multiply of the transforms: two packed products and a packed sum, dependent
            f2 w = b;
            if (KIND == 6) {                   // ... with the twiddle coming from LDS right before, as in the kernels
                lds[t] = b;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((unsigned)(t * 8)) : "memory");
            }
            f2 p, q;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p) : "v"(a), "v"(w));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(w));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(p));
            float p0, p1, q0, q1;
            asm volatile("v_mul_f32 %0, %1, -%2" : "=v"(p0) : "v"(a.y), "v"(w.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(a.y), "v"(w.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(a.x), "v"(w.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(a.x), "v"(w.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(rx) : "v"(q0), "v"(p0));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ry) : "v"(q1), "v"(p1));
        } else {                           // control: scalar against scalar
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rx) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.y));
        }
        nbad += (__float_as_uint(r.x) != __float_as_uint(rx)) | (__float_as_uint(r.y) != __float_as_uint(ry));
        // new operands every iteration, bounded
        a.x = 0.5f * rx + 0.7f; a.y = 0.25f * ry + 0.9f;
        b.x = b.x * 0.999f + 0.001f * (it & 15); b.y = 1.3f - 0.5f * b.x;
    }
    return nbad;
}

__device__ __forceinline__ float mfma_loop(int iters) {
    h8 a0, b0, a1, b1;
    for (int e = 0; e < 8; ++e) {
        a0[e] = (_Float16)(0.01f * ((threadIdx.x + e) & 31) - 0.1f); b0[e] = (_Float16)(0.02f * ((threadIdx.x * 3 + e) & 15) - 0.1f);
        a1[e] = (_Float16)(0.015f * ((threadIdx.x + 2 * e) & 31) - 0.2f); b1[e] = (_Float16)(0.01f * ((threadIdx.x * 5 + e) & 15));
    }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
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
    const f4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    return s.x + s.y + s.z + s.w;
}

template <int KIND>
__global__ __launch_bounds__(256) void victim_kernel(unsigned *bad_by_lane, int iters) {
    const unsigned n = victim_loop<KIND>(iters);
    if (n) atomicAdd(&bad_by_lane[threadIdx.x & 63], n);
}
__global__ __launch_bounds__(256) void aggressor_kernel(float *sink, int iters) {
    const float s = mfma_loop(iters);
    if (s == 12345.f) sink[0] = s;
}
template <int KIND>
__global__ __launch_bounds__(256) void mixed_kernel(unsigned *bad_by_lane, float *sink, int v_iters, int m_iters) {
    if (blockIdx.x & 1) {
        const float s = mfma_loop(m_iters);
        if (s == 12345.f) sink[0] = s;
    } else {
        const unsigned n = victim_loop<KIND>(v_iters);
        if (n) atomicAdd(&bad_by_lane[threadIdx.x & 63], n);
    }
}

static void report(const char *what, const unsigned *bad_dev, double checks) {
    unsigned h[64];
    hipMemcpy(h, bad_dev, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
    printf("  %-58s mismatches by lanes 0-15 / 16-31 / 32-47 / 48-63: %llu / %llu / %llu / %llu  (of %.3g checks)\n", what, q[0], q[1], q[2],
           q[3], checks);
    fflush(stdout);
}

template <int KIND>
static void legs(const char *name, unsigned *bad, float *sink, int cus) {
    const int v_wgs = cus * 4, v_iters = 40000, m_wgs = cus * 2, m_iters = 60000;
    const double checks = (double)v_wgs * 256 * v_iters;
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    printf("%s\n", name);
    hipMemset(bad, 0, 64 * 4);
    victim_kernel<KIND><<<v_wgs, 256, 0, s1>>>(bad, v_iters);
    hipDeviceSynchronize();
    report("alone", bad, checks);
    hipMemset(bad, 0, 64 * 4);
    aggressor_kernel<<<m_wgs, 256, 0, s2>>>(sink, m_iters);
    victim_kernel<KIND><<<v_wgs, 256, 0, s1>>>(bad, v_iters);
    hipDeviceSynchronize();
    report("beside v_mfma_f32_16x16x32_f16 waves of another stream", bad, checks);
    hipMemset(bad, 0, 64 * 4);
    mixed_kernel<KIND><<<v_wgs * 2, 256, 0, s1>>>(bad, sink, v_iters, m_iters / 2);
    hipDeviceSynchronize();
    report("beside MFMA workgroups of the SAME dispatch", bad, checks);
    hipStreamDestroy(s1); hipStreamDestroy(s2);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *bad; float *sink;
    hipMalloc(&bad, 64 * 4); hipMalloc(&sink, 64);
    legs<0>("v_pk_mul_f32", bad, sink, cus);
    legs<1>("v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]", bad, sink, cus);
    legs<2>("v_pk_fma_f32", bad, sink, cus);
    legs<3>("v_pk_add_f32", bad, sink, cus);
    legs<5>("complex multiply: 2 x v_pk_mul_f32 + v_pk_add_f32, dependent", bad, sink, cus);
    legs<6>("the same with the second operand read from LDS right before", bad, sink, cus);
    legs<4>("control: v_mul_f32 against v_mul_f32", bad, sink, cus);
    return 0;
}
