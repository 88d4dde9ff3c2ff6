// Spectral subtraction + ReLU + next ref_mag, fused, for gfx950.
//
// Replaces audio_complete.subtract (/root/reference/util_audio.py:221-259):
//     mag_sub  = guess.mag * (self.ref_mag / guess.ref_mag)   [normalize]
//     mag_sub *= overkill_factor
//     self.mag[:, off:off+Tg'] -= mag_sub[:, :Tg']             (Tg' clipped to the window end)
//     self.mag = max(self.mag, 0)                              [relu]
// and the np.max(self.mag) the reference re-evaluates on the next ref_mag read
// (the mag setter clears _ref_mag, util_audio.py:149-157, :170-174).
//
// HBM-bound elementwise pass over frame-major [T][ldf] windows: float4 (16 B
// per lane) loads/stores, one atomicMax per workgroup.  Arithmetic is kept
// un-fused (__fmul_rn / __fsub_rn, IEEE divide) so the residual is bit-identical
// to numpy's float32 result on the same inputs.
#include "amt_common.h"

__global__ __launch_bounds__(256) void subtract_kernel(amt_subtract_args a,
                                                        unsigned int *__restrict__ new_max_ord) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const int g = a.guess_index ? a.guess_index[b] : b;
    const int tg = a.guess_frames ? a.guess_frames[b] : a.guess_frames_all;
    int off = a.offset_frames ? a.offset_frames[b] : 0;
    if (off < 0) off = 0;
    int t_end = off + tg;                               // exclusive; clipped to T (:250-251)
    if (t_end > a.T) t_end = a.T;
    float scale = 1.0f;
    if (a.normalize) scale = __fdiv_rn(a.resid_max[b], a.guess_max[g]);
    const float overkill = a.overkill_factor;

    float4 *r4 = reinterpret_cast<float4 *>(a.resid + (size_t)b * a.resid_stride);
    const float4 *g4 = reinterpret_cast<const float4 *>(a.guess + (size_t)g * a.guess_stride);
    const int ld4 = a.ldf >> 2;
    const int n4 = a.T * ld4;
    float m = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const int t = i / ld4;
        const int f4 = i - t * ld4;
        float4 r = r4[i];
        if (t >= off && t < t_end) {
            const float4 q = g4[(size_t)(t - off) * ld4 + f4];
            r.x = __fsub_rn(r.x, __fmul_rn(__fmul_rn(q.x, scale), overkill));
            r.y = __fsub_rn(r.y, __fmul_rn(__fmul_rn(q.y, scale), overkill));
            r.z = __fsub_rn(r.z, __fmul_rn(__fmul_rn(q.z, scale), overkill));
            r.w = __fsub_rn(r.w, __fmul_rn(__fmul_rn(q.w, scale), overkill));
        }
        if (a.relu) {
            r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f);
            r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
        }
        r4[i] = r;
        const int f = f4 << 2;                          // pad columns stay out of the max
        if (f < a.F) m = fmaxf(m, r.x);
        if (f + 1 < a.F) m = fmaxf(m, r.y);
        if (f + 2 < a.F) m = fmaxf(m, r.z);
        if (f + 3 < a.F) m = fmaxf(m, r.w);
    }
    if (new_max_ord) {
        m = block_max(m, red);
        if (threadIdx.x == 0) atomicMax(new_max_ord + b, float_to_ordered(m));
    }
}

__global__ void sub_ordered_init_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = AMT_ORDERED_NEG_INF;
}
__global__ void sub_ordered_decode_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __float_as_uint(ordered_to_float(p[i]));
}

extern "C" int amt_subtract(const amt_subtract_args *args, void *stream) {
    if (!args || !args->resid || !args->guess) return AMT_E_INVALID;
    const amt_subtract_args &a = *args;
    if (a.B <= 0 || a.T <= 0 || a.F <= 0 || a.ldf < a.F || (a.ldf & 3)) return AMT_E_SHAPE;
    if ((a.resid_stride & 3) || (a.guess_stride & 3)) return AMT_E_SHAPE;
    if (a.resid_stride < (size_t)a.T * a.ldf) return AMT_E_SHAPE;
    if (a.normalize && (!a.resid_max || !a.guess_max)) return AMT_E_INVALID;
    if (!a.guess_frames && a.guess_frames_all < 0) return AMT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    unsigned int *nm = reinterpret_cast<unsigned int *>(a.new_max);
    if (nm) sub_ordered_init_kernel<<<(a.B + 255) / 256, 256, 0, st>>>(nm, a.B);
    const int n4 = a.T * (a.ldf >> 2);
    int gx = (n4 + 256 * 4 - 1) / (256 * 4);            // ~4 float4 per thread
    if (gx < 1) gx = 1;
    // cap the grid at ~16 workgroups per CU worth of blocks and grid-stride the rest
    const long cap = 4096;
    if ((long)gx * a.B > cap) { gx = (int)(cap / a.B); if (gx < 1) gx = 1; }
    subtract_kernel<<<dim3(gx, a.B), 256, 0, st>>>(a, nm);
    if (nm) sub_ordered_decode_kernel<<<(a.B + 255) / 256, 256, 0, st>>>(nm, a.B);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
