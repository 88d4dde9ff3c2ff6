// Sustained f16 MFMA rate of this chip with nothing else in the way: every wave issues
// v_mfma_f32_16x16x32_f16 back to back on register operands (no LDS, no memory).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s for zero operands and for random non-zero operands (the matrix pipe's power
// draw depends on the data; the clock the power limit allows sets the rate), with 1, 2 and 4 waves
// per SIMD.  The convolution's "executed MFMA" rate in bench.py is to be read against these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// the same with v_mfma_f32_32x32x16_f16 (twice the flops per operand byte read from the register file)
__global__ __launch_bounds__(256) void mfma_loop32(const h8 *__restrict__ a_in, const h8 *__restrict__ b_in, float *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a0 = a_in[t], b0 = b_in[t], a1 = a_in[t ^ 1], b1 = b_in[t ^ 1];
    f16v c0 = {0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
    }
    const f16v s = c0 + c1 + c2 + c3;
    float r = 0.f;
    for (int e = 0; e < 16; ++e) r += s[e];
    out[t] = r;
}

__global__ __launch_bounds__(256) void mfma_loop(const h8 *__restrict__ a_in, const h8 *__restrict__ b_in, float *out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    h8 a0 = a_in[t], b0 = b_in[t], a1 = a_in[t ^ 1], b1 = b_in[t ^ 1];
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
    out[t] = s.x + s.y + s.z + s.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    const int max_threads = cus * 4 * 256;
    std::vector<_Float16> hz(max_threads * 8, (_Float16)0.f), hr(max_threads * 8);
    srand(1);
    for (auto &v : hr) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    h8 *az, *ar;
    float *out;
    CK(hipMalloc(&az, max_threads * sizeof(h8)));
    CK(hipMalloc(&ar, max_threads * sizeof(h8)));
    CK(hipMalloc(&out, max_threads * sizeof(float)));
    CK(hipMemcpy(az, hz.data(), max_threads * sizeof(h8), hipMemcpyHostToDevice));
    CK(hipMemcpy(ar, hr.data(), max_threads * sizeof(h8), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400000;      // 25 - 100 ms per launch: long enough for the power management to settle
    for (int data = 0; data < 2; ++data) {
        for (int wps = 1; wps <= 4; wps *= 2) {             // waves per SIMD = workgroups of 4 waves per CU
            const int blocks = cus * wps;
            const h8 *src = data ? ar : az;
            mfma_loop<<<blocks, 256>>>(src, src, out, 2000);
            CK(hipDeviceSynchronize());
            float best = 0.f, sum = 0.f;
            const int reps = 3;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0));
                mfma_loop<<<blocks, 256>>>(src, src, out, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double fl = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
                const float tf = (float)(fl / (ms * 1e-3) / 1e12);
                best = tf > best ? tf : best;
                sum += tf;
            }
            printf("%-22s %d wave(s)/SIMD: %7.1f TFLOP/s mean, %7.1f best of %d launches of %.0f ms\n",
                   data ? "random operands" : "zero operands", wps, sum / reps, best, reps,
                   (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32) / (best * 1e12) * 1e3);
        }
    }
    for (int data = 0; data < 2; ++data) {
        const int wps = 4, blocks = cus * wps, it32 = iters;     // 4 x (2 x 32*32*16) per iteration = the same flops as 8 x 16x16x32
        const h8 *src = data ? ar : az;
        mfma_loop32<<<blocks, 256>>>(src, src, out, 2000);
        CK(hipDeviceSynchronize());
        float best = 0.f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            mfma_loop32<<<blocks, 256>>>(src, src, out, it32);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = (double)blocks * 4 * it32 * 4 * (2.0 * 32 * 32 * 16);
            const float tf = (float)(fl / (ms * 1e-3) / 1e12);
            best = tf > best ? tf : best;
        }
        printf("32x32x16: %-18s %d wave(s)/SIMD: %7.1f TFLOP/s best of 3\n", data ? "random operands" : "zero operands", wps, best);
    }
    return 0;
}
