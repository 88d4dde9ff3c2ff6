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

// ---- the same step on the frames that change ------------------------------------------------------------------------
// Only frames [off, t_end) of a window differ after the subtraction (util_audio.py:250-259 slices them out); when the
// residual is non-negative everywhere -- magnitudes, or the result of an earlier ReLU -- the ReLU leaves every other
// frame as it is, and np.max(self.mag) is the maximum of per-frame maxima of which only those frames' change.  With a
// per-frame maximum array kept beside the spectrogram (amt_compress_bands_fmax writes it while it has each frame in
// registers) the step reads and writes the guess's span instead of the window: ~2.0 instead of 4.9 MB per window and
// iteration at the loop's 173-frame guesses.  One wave per frame, float4 per lane, the same un-fused arithmetic as
// subtract_kernel: bit-identical residuals and maxima.
__global__ __launch_bounds__(256) void subtract_span_kernel(amt_subtract_args a, float *__restrict__ frame_max) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int g = a.guess_index ? a.guess_index[b] : b;
    const int tg = a.guess_frames ? a.guess_frames[b] : a.guess_frames_all;
    int off = a.offset_frames ? a.offset_frames[b] : 0;
    if (off < 0) off = 0;
    int t_end = off + tg;
    if (t_end > a.T) t_end = a.T;
    const int t = off + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= t_end) return;                                  // whole wave
    float scale = 1.0f;
    if (a.normalize) scale = __fdiv_rn(a.resid_max[b], a.guess_max[g]);
    const float overkill = a.overkill_factor;
    const int ld4 = a.ldf >> 2;
    float4 *r4 = reinterpret_cast<float4 *>(a.resid + (size_t)b * a.resid_stride) + (size_t)t * ld4;
    const float4 *g4 = reinterpret_cast<const float4 *>(a.guess + (size_t)g * a.guess_stride) + (size_t)(t - off) * ld4;
    float m = 0.f;
    for (int f4 = lane; f4 < ld4; f4 += 64) {
        float4 r = r4[f4];
        const float4 q = g4[f4];
        r.x = fmaxf(__fsub_rn(r.x, __fmul_rn(__fmul_rn(q.x, scale), overkill)), 0.f);
        r.y = fmaxf(__fsub_rn(r.y, __fmul_rn(__fmul_rn(q.y, scale), overkill)), 0.f);
        r.z = fmaxf(__fsub_rn(r.z, __fmul_rn(__fmul_rn(q.z, scale), overkill)), 0.f);
        r.w = fmaxf(__fsub_rn(r.w, __fmul_rn(__fmul_rn(q.w, scale), overkill)), 0.f);
        r4[f4] = r;
        const int f = f4 << 2;                               // pad columns stay out of the max
        if (f < a.F) m = fmaxf(m, r.x);
        if (f + 1 < a.F) m = fmaxf(m, r.y);
        if (f + 2 < a.F) m = fmaxf(m, r.z);
        if (f + 3 < a.F) m = fmaxf(m, r.w);
    }
    m = wave_max(m);
    if (lane == 0) frame_max[(size_t)b * a.T + t] = m;
}
__global__ __launch_bounds__(256) void frames_max_kernel(const float *__restrict__ frame_max, int T, float *__restrict__ out) {
    __shared__ float red[16];
    const float *row = frame_max + (size_t)blockIdx.x * T;
    float m = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, row[t]);
    m = block_max(m, red);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

__global__ void sub_ordered_init_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = AMT_ORDERED_NEG_INF;
}
__global__ void sub_ordered_decode_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __float_as_uint(ordered_to_float(p[i]));
}

extern "C" int amt_subtract_span(const amt_subtract_args *args, float *frame_max, int span_cap, void *stream) {
    if (!args || !args->resid || !args->guess || !frame_max) return AMT_E_INVALID;
    const amt_subtract_args &a = *args;
    if (a.B <= 0 || a.T <= 0 || a.F <= 0 || a.ldf < a.F || (a.ldf & 3)) return AMT_E_SHAPE;
    if ((a.resid_stride & 3) || (a.guess_stride & 3)) return AMT_E_SHAPE;
    if (a.resid_stride < (size_t)a.T * a.ldf) return AMT_E_SHAPE;
    if (!a.relu) return AMT_E_UNSUPPORTED;                    // the other frames are only untouched by a ReLU of >= 0 values
    if (a.normalize && (!a.resid_max || !a.guess_max)) return AMT_E_INVALID;
    if (!a.guess_frames && a.guess_frames_all < 0) return AMT_E_INVALID;
    if (span_cap <= 0) return AMT_E_INVALID;                  // >= every window's guess_frames (the guess tensor's frame count)
    // a uniform frame count beyond the launch's frame grid would be left unsubtracted (per-window counts live on the
    // device: their bound is the caller's contract, span_cap = the guess tensor's frame count)
    if (!a.guess_frames && a.guess_frames_all > span_cap) return AMT_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int cap = span_cap < a.T ? span_cap : a.T;
    subtract_span_kernel<<<dim3((cap + 3) / 4, a.B), 256, 0, st>>>(a, frame_max);
    if (a.new_max) frames_max_kernel<<<a.B, 256, 0, st>>>(frame_max, a.T, a.new_max);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
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
