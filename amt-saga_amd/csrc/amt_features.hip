// Feature gathers that turn the residual spectrogram into the RDCNN head inputs.
//
// Replaces, for a batch of windows at once:
//   audio_complete.compress_bands (util_audio.py:436-466) + _resize (:384-409)
//       -> C_timing of training.py:333-336
//   audio_complete.resize(..., ['mag','ph']) (:469-507) + section_power (:334-349)
//       + the scalings of training.py:347-363 -> F_sw_inst_foc* / ph features
// All HBM-bound gathers over frame-major spectra; outputs are laid out
// [B][bands][frames] = the NHWC (C = 1) tensors the heads consume.
#include "amt_common.h"
#include <algorithm>

// One wave per (window, output frame): the row is read coalesced (lane + 64*q) by 17 branch-free loads from clamped
// addresses, all in flight together.
//
// Fast path -- the loop's own configuration, F = 1025 bins into the 20 log-spaced bands of util_audio.py:451-456
// (edges 0 1 2 3 4 5 8 11 16 22 32 45 64 90 128 181 256 362 512 724 1025; checked against the caller's edge array by
// scalar compares, uniform): with the edges known at compile time every band is a fixed list of whole and partial
// registers -- no per-register skip branches (the generic form walks 17 uniform branches per band), masks only on the
// partial ones -- and its wave reduction runs on the vector units alone: DPP row rotations inside the four 16-lane
// rows, the four row totals by v_readlane.  One division and one store pass at the end (lane i holds band i).
// Generic path (any edges / F): narrow bands lane-per-band through wave shuffles, wide bands by masked sums + a
// shuffle reduction.  F <= 64*MAXQ.
template <int N>
__device__ __forceinline__ float cb_row_ror(float v) {                 // lane i of a 16-lane row <- lane (i + N) % 16
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float cb_wave_sum(float v) {
    v += cb_row_ror<8>(v);
    v += cb_row_ror<4>(v);
    v += cb_row_ror<2>(v);
    v += cb_row_ror<1>(v);                                             // every lane: the total of its 16-lane row
    const int u = __float_as_int(v);                                   // (the builtin moves integers)
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(u, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(u, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(u, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(u, 48));
    return (r0 + r1) + (r2 + r3);
}
#define CB_STD_BANDS 20
#define CB_STD_F 1025
__device__ constexpr int cb_std_edge(int i) {
    constexpr int e[CB_STD_BANDS + 1] = {0, 1, 2, 3, 4, 5, 8, 11, 16, 22, 32, 45, 64, 90, 128, 181, 256, 362, 512, 724, 1025};
    return e[i];
}
template <int I, int MAXQ>
__device__ __forceinline__ void cb_std_band(const float (&v)[MAXQ], int lane, float &mine) {
    constexpr int lo = cb_std_edge(I), hi = cb_std_edge(I + 1);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        if (hi <= 64 * q || lo >= 64 * (q + 1)) continue;              // compile time: the band does not touch this register
        if (lo <= 64 * q && hi >= 64 * (q + 1)) s += v[q];             // compile time: the whole register
        else { const int f = lane + 64 * q; s += (f >= lo && f < hi) ? v[q] : 0.f; }
    }
    s = cb_wave_sum(s);
    mine = lane == I ? s : mine;                            // a select: the reduction must run with every lane active
    if constexpr (I + 1 < CB_STD_BANDS) cb_std_band<I + 1, MAXQ>(v, lane, mine);
}
template <int MAXQ>
__global__ __launch_bounds__(256) void compress_bands_kernel(
    const float *__restrict__ mag, int T, int F, int ldf, size_t spec_stride,
    const int32_t *__restrict__ edges, int bands, const float *__restrict__ ref,
    const int32_t *__restrict__ src_frame, float *__restrict__ out, int target, float *__restrict__ frame_max) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // output frame
    const int b = blockIdx.y;
    if (j >= target) return;                                // whole wave exits together
    const int t = src_frame ? src_frame[j] : j;
    float v[MAXQ];
    const bool frame_ok = t >= 0 && t < T;
    const float *row = mag + (size_t)b * spec_stride + (size_t)(frame_ok ? t : 0) * ldf;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) v[q] = row[min(lane + 64 * q, F - 1)];
    const float r = ref ? ref[b] : 1.0f;
    float *o = out + ((size_t)b * bands) * target + j;
    bool std_edges = MAXQ == 17 && bands == CB_STD_BANDS && F == CB_STD_F;
    if (std_edges) {
#pragma unroll
        for (int i = 0; i <= CB_STD_BANDS; ++i) std_edges = std_edges && edges[i] == cb_std_edge(i);
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) v[q] = (frame_ok && lane + 64 * q < F) ? v[q] : 0.f;
    if (frame_max && frame_ok) {
        // the frame is in registers anyway: its maximum, for amt_subtract_span (identity frame map: every frame once)
        float m = v[0];
#pragma unroll
        for (int q = 1; q < MAXQ; ++q) m = fmaxf(m, v[q]);
        m = wave_max(m);
        if (lane == 0) frame_max[(size_t)b * T + t] = m;
    }
    if constexpr (MAXQ == 17) {
        if (std_edges) {                                    // uniform
            float mine = 0.f;
            cb_std_band<0, MAXQ>(v, lane, mine);
            if (lane < CB_STD_BANDS)
                o[(size_t)lane * target] = __fdiv_rn(__fdiv_rn(mine, (float)(edges[lane + 1] - edges[lane])), r);
            return;
        }
    }
    // ---- generic edges ------------------------------------------------------------------------------------------
    // Bands that lie inside the first 64 bins are summed lane-per-band: lane i walks its band through wave shuffles of the
    // first register -- one pass of max-width steps for all of them instead of a 64-lane reduction each.  Wider
    // bands: masked sums over the registers the band touches (uniform skips) + one wave reduction.
    const int my_lo = lane < bands ? edges[lane] : 0, my_hi = lane < bands ? edges[lane + 1] : 0;
    const bool narrow = lane < bands && my_hi <= 64;
    int wmax = narrow ? my_hi - my_lo : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
    float sn = 0.f;
    for (int k = 0; k < wmax; ++k) {
        const float x = __shfl(v[0], (my_lo + k) & 63, 64);
        if (narrow && my_lo + k < my_hi) sn += x;
    }
    if (narrow) o[(size_t)lane * target] = __fdiv_rn(__fdiv_rn(sn, (float)(my_hi - my_lo)), r);
    for (int i = 0; i < bands; ++i) {
        const int lo = edges[i], hi = edges[i + 1];
        if (hi <= 64) continue;                                  // done above (uniform)
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            if (hi <= 64 * q || lo >= 64 * (q + 1)) continue;    // uniform: the band does not touch this register
            const int f = lane + 64 * q;
            s += (f >= lo && f < hi) ? v[q] : 0.f;
        }
        s = wave_sum(s);
        if (lane == 0) o[(size_t)i * target] = __fdiv_rn(__fdiv_rn(s, (float)(hi - lo)), r);
    }
}

// One workgroup per window: gather [bands][frames] from `frames` source rows.
__global__ __launch_bounds__(256) void short_window_kernel(
    const float *__restrict__ mag, const float2 *__restrict__ phase, int T, int F, int ldf,
    size_t spec_stride, const int32_t *__restrict__ src_frame, int frames,
    const int32_t *__restrict__ band_min, int bands, const float *__restrict__ ref, int mode,
    float *__restrict__ out) {
    __shared__ float red[16];
    __shared__ float bmax;
    const int b = blockIdx.x;
    const int n = bands * frames;
    const int lo = band_min ? band_min[b] : 0;
    const float *mg = mag + (size_t)b * spec_stride;
    const float2 *ph = phase ? phase + (size_t)b * spec_stride : nullptr;
    float *o = out + (size_t)b * n;
    float lmax = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / frames, j = i - r * frames;
        const int t = src_frame[b * frames + j];
        const int f = lo + r;
        const bool ok = t >= 0 && t < T && f >= 0 && f < F;
        float val;
        if (mode == 2) {
            const float2 q = ok ? ph[(size_t)t * ldf + f] : make_float2(0.f, 0.f);
            val = __fdiv_rn(atan2f(q.y, q.x) + 3.15f, 6.3f);
        } else {
            const float m = ok ? mg[(size_t)t * ldf + f] : 0.f;
            if (mode == 0) val = ref ? __fdiv_rn(m, ref[b]) : m;
            else { val = log10f(__fmul_rn(m, 1000.0f) + 1.0f); lmax = fmaxf(lmax, val); }
        }
        o[i] = val;
    }
    if (mode == 1) {
        lmax = block_max(lmax, red);
        if (threadIdx.x == 0) bmax = lmax;
        __syncthreads();
        const float d = bmax;
        for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = __fdiv_rn(o[i], d);
    }
}

// ---- window management as index maps ---------------------------------------------------------
// audio_complete.section / slice / concat / resize / section_power (util_audio.py:286-382, 469-507) all
// select frames (columns of the reference's [F][T] arrays = ROWS of the frame-major device layout) and,
// for section_power, a band of bins, padding with zeros past the end.  One gather kernel serves them:
//   out[b][j][k] = src[b][src_frame[j]][band_min + k]   (0 when src_frame[j] < 0 or >= T, or the bin >= F)
// `elem` floats per bin (1: magnitude / dB, 2: unit phase or complex F).  Whole rows move as 16-byte
// accesses when band_min = 0 and the pitches allow it, else element-wise; HBM-bound copies.
__global__ __launch_bounds__(256) void gather_frames_kernel(
    const float *__restrict__ src, int T, int F, int ldf_src, size_t src_stride, int elem,
    const int32_t *__restrict__ src_frame, int table_stride, int n_out, int band_min, int bands,
    float *__restrict__ out, int ldf_out, size_t out_stride) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int t = src_frame[(size_t)b * table_stride + j];
    const bool row_ok = t >= 0 && t < T;
    const float *s = src + (size_t)b * src_stride + (size_t)(row_ok ? t : 0) * ldf_src * elem;
    float *o = out + (size_t)b * out_stride + (size_t)j * ldf_out * elem;
    const int n = ldf_out * elem;                       // floats of the output row (pad bins written as zero)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int k = i / elem, c = i - k * elem;
        const int f = band_min + k;
        o[i] = (row_ok && k < bands && f >= 0 && f < F) ? s[(size_t)f * elem + c] : 0.f;
    }
}

// ---- amplitude <-> dB (audio_complete.D, util_audio.py:176-190; librosa.amplitude_to_db / db_to_amplitude) ----
// D = 20 log10(max(amin, |S|)) - 20 log10(max(amin, ref)), floored at max(D) - top_db.  log10 is monotone, so
// max(D) follows from the window's max |S| (amt_window_max / the fused STFT max: wave -> block -> atomic
// reductions), and the whole map is one elementwise pass.  Pad bins (f >= F) are written as zero.
__global__ __launch_bounds__(256) void amplitude_to_db_kernel(
    const float *__restrict__ mag, int T, int F, int ldf, size_t spec_stride,
    const float *__restrict__ ref, const float *__restrict__ wmax, float amin, float top_db,
    float *__restrict__ out) {
    const int b = blockIdx.y;
    const float lref = 20.0f * log10f(fmaxf(amin, fabsf(ref[b])));
    const float floor_db = top_db >= 0.f ? (20.0f * log10f(fmaxf(amin, wmax[b])) - lref) - top_db : -INFINITY;
    const size_t n = (size_t)T * ldf;
    const float *m = mag + (size_t)b * spec_stride;
    float *o = out + (size_t)b * spec_stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int f = (int)(i % ldf);
        float v = 0.f;
        if (f < F) v = fmaxf(20.0f * log10f(fmaxf(amin, fabsf(m[i]))) - lref, floor_db);
        o[i] = v;
    }
}
__global__ __launch_bounds__(256) void db_to_amplitude_kernel(
    const float *__restrict__ db, int T, int F, int ldf, size_t spec_stride, const float *__restrict__ ref,
    float *__restrict__ out) {
    const int b = blockIdx.y;
    const float r = ref[b];
    const size_t n = (size_t)T * ldf;
    const float *d = db + (size_t)b * spec_stride;
    float *o = out + (size_t)b * spec_stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int f = (int)(i % ldf);
        o[i] = f < F ? r * exp10f(0.05f * d[i]) : 0.f;
    }
}

// ---- spectral flatness per frame (audio_complete.spectral_flatness, util_audio.py:330-332 ->
// librosa.feature.spectral_flatness, power = 2): exp(mean(log(max(amin, S^2)))) / mean(max(amin, S^2)) over the
// F bins of a frame.  One wave per frame, coalesced row read, two wave reductions.
__global__ __launch_bounds__(256) void spectral_flatness_kernel(
    const float *__restrict__ mag, int T, int F, int ldf, size_t spec_stride, float amin,
    float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (t >= T) return;                                      // whole wave exits together
    const float *row = mag + (size_t)b * spec_stride + (size_t)t * ldf;
    float sl = 0.f, sa = 0.f;
    for (int f = lane; f < F; f += 64) {
        const float m = row[f];
        const float pw = fmaxf(amin, m * m);
        sl += logf(pw);
        sa += pw;
    }
    sl = wave_sum(sl); sa = wave_sum(sa);
    if (lane == 0) out[(size_t)b * T + t] = __fdiv_rn(expf(sl / (float)F), sa / (float)F);
}

extern "C" {

int amt_compress_bands_fmax(const float *mag, int B, int T, int F, int ldf, size_t spec_stride,
                            const int32_t *edges, int bands, const float *ref,
                            const int32_t *src_frame, float *out, int target_frames, float *frame_max, void *stream) {
    if (!mag || !edges || !out || B <= 0 || T <= 0 || bands <= 0 || target_frames <= 0)
        return AMT_E_INVALID;
    if (F <= 0 || ldf < F) return AMT_E_SHAPE;
    if (frame_max && (src_frame || target_frames != T)) return AMT_E_INVALID;       // every frame exactly once
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((target_frames + 3) / 4, B);
    if (F <= 64 * 5)
        compress_bands_kernel<5><<<grid, 256, 0, st>>>(mag, T, F, ldf, spec_stride, edges, bands, ref, src_frame, out, target_frames, frame_max);
    else if (F <= 64 * 17)
        compress_bands_kernel<17><<<grid, 256, 0, st>>>(mag, T, F, ldf, spec_stride, edges, bands, ref, src_frame, out, target_frames, frame_max);
    else if (F <= 64 * 33)
        compress_bands_kernel<33><<<grid, 256, 0, st>>>(mag, T, F, ldf, spec_stride, edges, bands, ref, src_frame, out, target_frames, frame_max);
    else
        return AMT_E_UNSUPPORTED;
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
int amt_compress_bands(const float *mag, int B, int T, int F, int ldf, size_t spec_stride,
                       const int32_t *edges, int bands, const float *ref,
                       const int32_t *src_frame, float *out, int target_frames, void *stream) {
    return amt_compress_bands_fmax(mag, B, T, F, ldf, spec_stride, edges, bands, ref, src_frame, out, target_frames, nullptr,
                                   stream);
}

int amt_short_window(const float *mag, const float *phase_ri, int B, int T, int F, int ldf,
                     size_t spec_stride, const int32_t *src_frame, int frames,
                     const int32_t *band_min, int bands, const float *ref, int mode, float *out,
                     void *stream) {
    if (!out || !src_frame || B <= 0 || T <= 0 || frames <= 0 || bands <= 0) return AMT_E_INVALID;
    if (mode < 0 || mode > 2) return AMT_E_INVALID;
    if ((mode == 2 && !phase_ri) || (mode != 2 && !mag)) return AMT_E_ATTRIB;
    if (F <= 0 || ldf < F) return AMT_E_SHAPE;
    short_window_kernel<<<B, 256, 0, (hipStream_t)stream>>>(
        mag, reinterpret_cast<const float2 *>(phase_ri), T, F, ldf, spec_stride, src_frame, frames,
        band_min, bands, ref, mode, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_gather_frames(const float *src, int B, int T, int F, int ldf_src, size_t src_stride, int elem,
                      const int32_t *src_frame, int table_stride, int n_out, int band_min, int bands,
                      float *out, int ldf_out, size_t out_stride, void *stream) {
    if (!src || !src_frame || !out || B <= 0 || T < 0 || n_out <= 0) return AMT_E_INVALID;
    if (elem < 1 || elem > 2) return AMT_E_ATTRIB;
    if (F <= 0 || ldf_src < F || bands <= 0 || ldf_out < bands) return AMT_E_SHAPE;
    if (table_stride != 0 && table_stride < n_out) return AMT_E_SHAPE;
    gather_frames_kernel<<<dim3(n_out, B), 256, 0, (hipStream_t)stream>>>(
        src, T, F, ldf_src, src_stride, elem, src_frame, table_stride, n_out, band_min, bands, out, ldf_out,
        out_stride);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_amplitude_to_db(const float *mag, int B, int T, int F, int ldf, size_t spec_stride, const float *ref,
                        const float *window_max, float amin, float top_db, float *out_db, void *stream) {
    if (!mag || !ref || !window_max || !out_db || B <= 0 || T <= 0) return AMT_E_INVALID;
    if (F <= 0 || ldf < F || !(amin > 0.f)) return AMT_E_SHAPE;
    const size_t n = (size_t)T * ldf;
    const unsigned gx = (unsigned)std::min<size_t>((n + 1023) / 1024, 1024);
    amplitude_to_db_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(mag, T, F, ldf, spec_stride, ref,
                                                                        window_max, amin, top_db, out_db);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_db_to_amplitude(const float *db, int B, int T, int F, int ldf, size_t spec_stride, const float *ref,
                        float *out_mag, void *stream) {
    if (!db || !ref || !out_mag || B <= 0 || T <= 0) return AMT_E_INVALID;
    if (F <= 0 || ldf < F) return AMT_E_SHAPE;
    const size_t n = (size_t)T * ldf;
    const unsigned gx = (unsigned)std::min<size_t>((n + 1023) / 1024, 1024);
    db_to_amplitude_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(db, T, F, ldf, spec_stride, ref, out_mag);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_spectral_flatness(const float *mag, int B, int T, int F, int ldf, size_t spec_stride, float amin,
                          float *out, void *stream) {
    if (!mag || !out || B <= 0 || T <= 0) return AMT_E_INVALID;
    if (F <= 0 || ldf < F || !(amin > 0.f)) return AMT_E_SHAPE;
    spectral_flatness_kernel<<<dim3((T + 3) / 4, B), 256, 0, (hipStream_t)stream>>>(mag, T, F, ldf, spec_stride,
                                                                                     amin, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

}  // extern "C"
