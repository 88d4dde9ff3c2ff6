// Constant-Q transform of the build's own spec (oracle/cqt.py) for gfx950: the <= 8 frames a head keeps
// (slice_C, /root/reference/util_audio.py:411-434: |librosa.cqt| then _resize(C[:, s:t], target)) and the
// maximum of the whole transform (the song-level normalisers of training.py:271-282).
//
// The transform: direct constant-Q response with periodic-Hann, L1-normalised filters of length N_k, scaled by
// sqrt(N_k); frequencies quantised to uint32 cycles/sample so the oscillator phase is exact integer
// arithmetic on CPU and GPU alike.  With w(n) = 1/2 - 1/2 cos(theta n), theta = 2 pi / N_k, a frame that
// starts at sample s is
//     C = 1/2 S0 - 1/4 ( e^{-i theta (s + N_k/2)} S+  +  e^{+i theta (s + N_k/2)} S- ),
//     S0 = sum x[m] e^{-i phi m},  S+- = sum x[m] e^{-i phi m} e^{+-i theta u},  u = m + N_k/2
// summed over the frame's support; frames start whole hops apart, so their supports are runs of the same
// H-sample blocks and every frame is a difference of two prefix sums of block sums plus one block head.
// VALU / L1-bound; no MFMA.
#include "amt_common.h"

#define AMT_CQT_MAXF 8

// ---------------------------------------------------------------------------------------------
// One kernel, two uses: the maximum over every bin and EVERY frame of a window's CQT (song-level
// normalisers, np.max(mid_wf.slice_C(0, duration, n_frames, ...)), training.py:271-282) and the <= 8
// frames of a slice (slice_C).
//
// One workgroup per (bin, window); the cost per bin is O(samples under the requested frames) whatever N_k is.  With block j =
// samples [jH - N_k/2, (j+1)H - N_k/2) (frame t starts at block t), the per-block sums
//     F_j = sum over the block,  G_j = sum over its first N_k mod H samples
// of (z, z e^{+i theta u}, z e^{-i theta u}), z = x[m] e^{-i phi m}, u = m + N_k/2, give every frame as
//     S_t = PF(t + nb) - PF(t) + G_{t+nb},   PF(j) = sum_{i<j} F_i (f64),   nb = N_k / H.
// Only the <= L/H + 2 blocks that hold samples are summed (PF is constant outside them), so the
// main loop is the same ~L/2048 passes per wave for a 376-sample filter and for an 887k-sample one.
// Pass: a wave stages 2048 consecutive samples through LDS (coalesced reads, prefetched one pass
// ahead in registers) and every lane takes 32 consecutive ones.  Inside such a segment the three
// phasors are  (phasor at the segment's first sample) x (e^{-i phi i}, e^{-i phi i +- i theta i}),
// i = 0..31: the second factor depends on the bin only, so it comes from a per-bin table through
// the scalar cache and a sample costs three packed FMAs -- no transcendental, no rotation -- and
// the exact-integer start phase is applied once per segment.  The running sums are snapshot after
// N_k mod 32 samples (the same i for every lane of the bin), which makes G_j a sum of whole segments
// and one snapshot: the prefix costs nothing per sample.  The waves never synchronise until the
// epilogue: a wave-parallel f64 scan of the F_j, then one thread per frame.
// ---------------------------------------------------------------------------------------------
#define CM_STAGE (64 * 36)              // floats per wave: 64 lanes x (32 samples + 4 pad): 16-B accesses, conflict-free
#define CM_COEF 192                     // floats per bin: 32 x (d0, d+, d-) complex

__global__ void cqt_coef_kernel(const unsigned int *__restrict__ phase_inc, const int *__restrict__ length, int n_bins,
                                float *__restrict__ coef) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_bins * 32) return;
    const int k = g >> 5, i = g & 31;
    const double ph = (double)phase_inc[k] * 4.656612873077393e-10 * (double)i;    // phi i in half-turns
    const double th = 2.0 / (double)length[k] * (double)i;                         // theta i
    float *o = coef + (size_t)k * CM_COEF + i * 6;
    double s, c;
    sincospi(-ph, &s, &c);      o[0] = (float)c; o[1] = (float)s;
    sincospi(th - ph, &s, &c);  o[2] = (float)c; o[3] = (float)s;
    sincospi(-th - ph, &s, &c); o[4] = (float)c; o[5] = (float)s;
}

struct cm_acc { amt_v2 s0, sp, sm; };
__device__ __forceinline__ void cm_zero(cm_acc &a) { a.s0 = amt_v2{0, 0}; a.sp = amt_v2{0, 0}; a.sm = amt_v2{0, 0}; }
__device__ __forceinline__ amt_v2 cm_mul(amt_v2 a, amt_v2 b) {         // complex product
    return __builtin_elementwise_fma(amt_v2{a.x, a.x}, b, amt_v2{a.y, a.y} * amt_v2{-b.y, b.x});
}

template <int N>
__device__ __forceinline__ float cm_row_ror(float v) {                 // lane i of a 16-lane row <- lane (i + N) % 16
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}

// f += sum_i xs[i] (d0, d+, d-)_i over eight samples; h = the sums before sample R (R < 0: no snapshot).
// R is a template parameter: as a run-time value the snapshot turns into six selects per sample, or into a
// branch per sample that serialises the scalar loads of the table.
template <int R>
__device__ __forceinline__ void cm_group(const float *xs, const amt_v2 *__restrict__ cf, cm_acc &f, cm_acc &h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i == R) h = f;
        const amt_v2 xx = amt_v2{xs[i], xs[i]};
        f.s0 = __builtin_elementwise_fma(xx, cf[3 * i], f.s0);
        f.sp = __builtin_elementwise_fma(xx, cf[3 * i + 1], f.sp);
        f.sm = __builtin_elementwise_fma(xx, cf[3 * i + 2], f.sm);
    }
}

struct CqtBlocksArgs {
    size_t wave_stride;
    int L, H, hshift, T;
    int blk_cap;
    unsigned int *out_max;      // max mode: [B] float bits, zeroed by the caller
    const int *src_frame;       // slices mode: [B][frames], -1 => zero column
    const int *bin0;            // slices mode: [B] first table row of the window, or null
    const float *ref;           // slices mode: [B] divisor or null
    float *out;                 // slices mode: [B][n_bins][frames]
    int frames, n_bins, n_table;
    float *fg_ws;               // GBUF: [workgroups][blk_cap][12] block sums in HBM (signals too long for the LDS)
    double *pf_ws;              // GBUF: [workgroups][blk_cap + 1][6]
};

// SLICES = false: every frame 0 .. T-1, maximum over frames -> atomicMax(out_max[b])   (amt_cqt_window_max)
// SLICES = true:  the <= 8 frames src_frame[b][*] -> out[b][k][*] / ref[b]              (amt_cqt_slices); only the
//                 blocks those frames touch are summed
// (the read-only arrays are separate __restrict__ parameters: as struct members the table reads lose the
// no-alias guarantee and stop being scalar loads)
// GBUF = true keeps the block sums and their prefix sums in an HBM workspace instead of the LDS: a whole song as one
// "window" (the reference's own normalisers are maxima over the song's CQT) has tens of thousands of hop-blocks.
template <bool SLICES, bool GBUF>
__global__ __launch_bounds__(256) void cqt_blocks_kernel(const float *__restrict__ wave,
                                                         const unsigned int *__restrict__ phase_inc,
                                                         const int *__restrict__ length,
                                                         const float *__restrict__ coef, CqtBlocksArgs a) {
    extern __shared__ double cm_smem[];
    float *stage = (float *)cm_smem;                 // [4][CM_STAGE]; the LDS form's epilogue reuses it for PF
    const int k = blockIdx.x, b = blockIdx.y;
    const size_t wg = (size_t)b * gridDim.x + k;
    double *pf = GBUF ? a.pf_ws + wg * ((size_t)a.blk_cap + 1) * 6 : cm_smem;                        // [nblk + 1][6]
    float *fg = GBUF ? a.fg_ws + wg * (size_t)a.blk_cap * 12
                     : stage + max(4 * CM_STAGE, (a.blk_cap + 1) * 12);                               // [blk_cap][12]  (F_j, G_j)
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int L = a.L, H = a.H, hshift = a.hshift, T = a.T;
    int kt = k, t_min = 0, t_max = T - 1;
    if (SLICES) {
        kt += a.bin0 ? __builtin_amdgcn_readfirstlane(a.bin0[b]) : 0;
        t_min = 0x7fffffff; t_max = -1;
        for (int j = 0; j < a.frames; ++j) {
            const int t = a.src_frame[b * a.frames + j];
            if (t >= 0) { t_min = min(t_min, t); t_max = max(t_max, t); }
        }
        if (kt < 0 || kt >= a.n_table || t_max < 0) {       // uniform: outside the table / an empty slice -> zeros
            if (tid < a.frames) a.out[((size_t)b * a.n_bins + k) * a.frames + tid] = 0.f;
            return;
        }
        t_min = __builtin_amdgcn_readfirstlane(t_min);
        t_max = __builtin_amdgcn_readfirstlane(t_max);
    }
    const int nk = __builtin_amdgcn_readfirstlane(length[kt]);
    const unsigned int inc = __builtin_amdgcn_readfirstlane(phase_inc[kt]);
    const amt_v2 *__restrict__ cf = (const amt_v2 *)(coef + (size_t)kt * CM_COEF);
    const float *x = wave + (size_t)b * a.wave_stride;
    const int half = nk >> 1;
    const int nb = nk >> hshift, rpre = nk - (nb << hshift);
    const int LPB = H >> 5;                          // lanes per block
    const int BPW = 64 / LPB, NBP = 4 * BPW;         // blocks per wave / per pass of the workgroup
    // blocks that hold samples AND lie under a requested frame (frame t = blocks t .. t+nb)
    const int j_lo = max(half >> hshift, t_min), j_hi = min((L - 1 + half) >> hshift, t_max + nb);
    const int nblk = max(j_hi - j_lo + 1, 0);        // <= blk_cap
    const float inv_nk = 1.0f / (float)nk;
    const int r32 = rpre & 31, qcut = rpre >> 5;     // G_j = segments < qcut whole + the first r32 samples of segment qcut
    const int gq = r32 >> 3, r8 = r32 & 7;
    const int seg = lane & (LPB - 1);                // this lane's segment inside its block

    const int n_pass = (nblk + NBP - 1) / NBP;
    typedef float cm_f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-B load, dword-aligned address
    typedef float cm_f4 __attribute__((ext_vector_type(4)));
    cm_f4 pre[8];
    auto fetch = [&](int c) {                        // the wave's 2048 samples of pass c -> registers
        const int m0 = ((j_lo + c * NBP + wid * BPW) << hshift) - half;
        if (m0 >= 0 && m0 + 2048 <= L) {             // wave-uniform; 1 KiB contiguous per instruction
            const float *xb = x + m0 + 4 * lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) pre[q] = *(const cm_f4u *)(xb + 256 * q);
        } else {                                     // the run hangs over an end of the window: zeros outside
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + 256 * q + 4 * lane + e;
                    const float xv = x[min(max(m, 0), L - 1)];
                    v[e] = (m >= 0 && m < L) ? xv : 0.f;
                }
                pre[q] = cm_f4{v[0], v[1], v[2], v[3]};
            }
        }
    };
    if (n_pass > 0) fetch(0);
    int wrem = ((j_lo << hshift) + (wid * BPW << hshift) + lane * 32) % nk;     // (ms + N_k/2) mod N_k, kept incrementally
    const int wstep = (NBP << hshift) % nk;
    for (int c = 0; c < n_pass; ++c) {
        const int jw = j_lo + c * NBP + wid * BPW;   // first block of this wave
        const int m0 = (jw << hshift) - half;
        float *sw_ = stage + wid * CM_STAGE;
        // sample s = 256 q + 4 lane + e of the run belongs to segment s / 32 = 8 q + lane / 8, place s % 32 = 4 (lane % 8) + e
#pragma unroll
        for (int q = 0; q < 8; ++q) *(cm_f4 *)(sw_ + (8 * q + (lane >> 3)) * 36 + 4 * (lane & 7)) = pre[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float *sl = sw_ + lane * 36;
        float xs[32];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const cm_f4 v = *(const cm_f4 *)(sl + 4 * i);
            xs[4 * i] = v.x; xs[4 * i + 1] = v.y; xs[4 * i + 2] = v.z; xs[4 * i + 3] = v.w;
        }
        if (c + 1 < n_pass) fetch(c + 1);            // in flight under the arithmetic below
        cm_acc f, h, g;
        cm_zero(f); cm_zero(h); cm_zero(g);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            if (gi == gq) {                          // uniform: the group that holds the snapshot position
                switch (r8) {
#define CM_CASE(R) case R: cm_group<R>(xs + 8 * gi, cf + 24 * gi, f, h); break;
                    CM_CASE(0) CM_CASE(1) CM_CASE(2) CM_CASE(3) CM_CASE(4) CM_CASE(5) CM_CASE(6) CM_CASE(7)
#undef CM_CASE
                }
            } else {
                cm_group<-1>(xs + 8 * gi, cf + 24 * gi, f, h);
            }
        }
        if (seg < qcut) g = f;
        else if (seg == qcut) g = h;
        // the segment's start phases: e^{-i phi ms} and e^{+-i theta (ms + N_k/2)}, integer-exact
        const int ms = m0 + lane * 32;               // ms + half >= 0
        const float turns = (float)((unsigned int)ms * inc) * 2.3283064365386963e-10f;
        const amt_v2 osc = amt_v2{__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns)};
        const float wt = (float)wrem * inv_nk;
        wrem += wstep;                               // (ms + N_k/2) mod N_k of the next pass
        wrem -= wrem >= nk ? nk : 0;
        const amt_v2 wp = amt_v2{__builtin_amdgcn_cosf(wt), __builtin_amdgcn_sinf(wt)};
        const amt_v2 op = cm_mul(osc, wp), om = cm_mul(osc, amt_v2{wp.x, -wp.y});
        f.s0 = cm_mul(f.s0, osc); f.sp = cm_mul(f.sp, op); f.sm = cm_mul(f.sm, om);
        g.s0 = cm_mul(g.s0, osc); g.sp = cm_mul(g.sp, op); g.sm = cm_mul(g.sm, om);
        float red[12] = {f.s0.x, f.s0.y, f.sp.x, f.sp.y, f.sm.x, f.sm.y, g.s0.x, g.s0.y, g.sp.x, g.sp.y, g.sm.x, g.sm.y};
        if (LPB >= 16) {                             // rows of 16 lanes by DPP rotation (VALU), the rest by permute
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                red[e] += cm_row_ror<8>(red[e]);
                red[e] += cm_row_ror<4>(red[e]);
                red[e] += cm_row_ror<2>(red[e]);
                red[e] += cm_row_ror<1>(red[e]);
            }
            for (int off = 16; off < LPB; off <<= 1) {
#pragma unroll
                for (int e = 0; e < 12; ++e) red[e] += __shfl_xor(red[e], off, 64);
            }
        } else {
            for (int off = 1; off < LPB; off <<= 1) {
#pragma unroll
                for (int e = 0; e < 12; ++e) red[e] += __shfl_xor(red[e], off, 64);
            }
        }
        const int jj = jw - j_lo + lane / LPB;       // block index among the blocks that hold samples
        if (seg == 0 && jj < nblk) {
            float *o = fg + (size_t)jj * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) o[e] = red[e];
        }
    }
    __syncthreads();                                 // every F_j, G_j is in LDS; the staging area is free

    // PF[i] = sum_{jj < i} F_jj in f64.  32 lanes per component: a lane sums its run of blocks, the 32 run totals
    // are scanned by shuffles, the lane writes its run's prefixes.
    if (tid < 192) {
        const int e = tid >> 5, l = tid & 31;
        const int per = (nblk + 31) >> 5;
        const int i0 = min(l * per, nblk), i1 = min(i0 + per, nblk);
        double sum = 0.0;
        for (int i = i0; i < i1; ++i) sum += (double)fg[(size_t)i * 12 + e];
        double v = sum;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const double u = __shfl_up(v, off, 32);
            if (l >= off) v += u;
        }
        double run = v - sum;                        // blocks before this lane's run
        if (l == 0) pf[e] = 0.0;
        for (int i = i0; i < i1; ++i) {
            run += (double)fg[(size_t)i * 12 + e];
            pf[(size_t)(i + 1) * 6 + e] = run;
        }
    }
    __syncthreads();
    const float scale = 2.0f / sqrtf((float)nk);
    auto frame = [&](int t) -> float {               // frame t: blocks t .. t+nb-1 whole, the head of block t+nb
        const int i0 = min(max(t - j_lo, 0), nblk), i1 = min(max(t + nb - j_lo, 0), nblk);
        const int jg = t + nb - j_lo;
        float sfr[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const float gv = (jg >= 0 && jg < nblk) ? fg[(size_t)jg * 12 + 6 + e] : 0.f;
            sfr[e] = (float)(pf[(size_t)i1 * 6 + e] - pf[(size_t)i0 * 6 + e]) + gv;
        }
        const float dt = (float)((t << hshift) % nk) * inv_nk;        // theta t H in turns (t H <= L)
        const float sd = __builtin_amdgcn_sinf(dt), cd = __builtin_amdgcn_cosf(dt);
        const float ar = cd * sfr[2] + sd * sfr[3], ai = cd * sfr[3] - sd * sfr[2];     // e^{-i d} S+
        const float br = cd * sfr[4] - sd * sfr[5], bi = cd * sfr[5] + sd * sfr[4];     // e^{+i d} S-
        const float re = 0.5f * sfr[0] - 0.25f * (ar + br);
        const float im = 0.5f * sfr[1] - 0.25f * (ai + bi);
        return sqrtf(re * re + im * im) * scale;
    };
    if (SLICES) {
        if (tid < a.frames) {
            const int t = a.src_frame[b * a.frames + tid];
            float v = t >= 0 ? frame(t) : 0.f;
            if (a.ref) v = __fdiv_rn(v, a.ref[b]);
            a.out[((size_t)b * a.n_bins + k) * a.frames + tid] = v;
        }
        return;
    }
    float vmax = 0.f;
    for (int t = tid; t < T; t += 256) vmax = fmaxf(vmax, frame(t));
    vmax = wave_max(vmax);
    if (lane == 0) atomicMax(a.out_max + b, __float_as_uint(vmax));   // non-negative floats order as their bits
}

// Fallback for signals whose block sums do not fit the LDS (a whole song handed to slice_C): every requested
// frame summed directly over its own N_k samples, one workgroup per (bin, window).  Any L; 8 N_k sample visits.
__global__ __launch_bounds__(256) void cqt_slices_direct_kernel(amt_cqt_args a) {
    __shared__ float red[4][2];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int kt = (a.bin0 ? a.bin0[b] : 0) + k;
    float *o = a.out + ((size_t)b * a.n_bins + k) * a.frames;
    if (kt < 0 || kt >= a.n_table) {                // uniform
        if (tid < a.frames) o[tid] = 0.f;
        return;
    }
    const int nk = a.length[kt];
    const unsigned int inc = a.phase_inc[kt];
    const float *x = a.wave + (size_t)b * a.wave_stride;
    const float inv_nk = 1.0f / (float)nk, scale = 2.0f / sqrtf((float)nk);
    for (int j = 0; j < a.frames; ++j) {
        const int t = a.src_frame[b * a.frames + j];
        float re = 0.f, im = 0.f;
        if (t >= 0) {
            const long long s0 = (long long)t * a.hop - (nk >> 1);
            for (int n = tid; n < nk; n += 256) {
                const long long m = s0 + n;
                if (m < 0 || m >= a.L) continue;
                const float xv = x[m];
                const float turns = (float)((unsigned int)m * inc) * 2.3283064365386963e-10f;
                const float w = 0.5f - 0.5f * __builtin_amdgcn_cosf((float)n * inv_nk);
                re += xv * w * __builtin_amdgcn_cosf(turns);
                im -= xv * w * __builtin_amdgcn_sinf(turns);
            }
        }
        re = wave_sum(re); im = wave_sum(im);
        __syncthreads();
        if (lane == 0) { red[wid][0] = re; red[wid][1] = im; }
        __syncthreads();
        if (tid == 0) {
            const float r = red[0][0] + red[1][0] + red[2][0] + red[3][0];
            const float i = red[0][1] + red[1][1] + red[2][1] + red[3][1];
            float v = t >= 0 ? sqrtf(r * r + i * i) * scale : 0.f;
            if (a.ref) v = __fdiv_rn(v, a.ref[b]);
            o[j] = v;
        }
    }
}

static int cqt_blocks_geometry(int L, int hop, int *hshift, int *blk_cap, size_t *lds) {
    if (!amt_is_pow2(hop) || hop < 128 || hop > 2048) return AMT_E_UNSUPPORTED;
    *hshift = 0;
    while ((1 << *hshift) < hop) ++*hshift;
    *blk_cap = L / hop + 3;                          // blocks that can hold samples, whatever the filter length
    const size_t stage_floats = (size_t)4 * CM_STAGE > (size_t)(*blk_cap + 1) * 12 ? (size_t)4 * CM_STAGE
                                                                                    : (size_t)(*blk_cap + 1) * 12;
    *lds = (stage_floats + (size_t)*blk_cap * 12) * sizeof(float);
    return *lds > 159 * 1024 ? AMT_E_UNSUPPORTED : AMT_OK;
}

extern "C" int amt_cqt_coef(const uint32_t *phase_inc, const int32_t *length, int n_table, float *coef, void *stream) {
    if (!phase_inc || !length || !coef || n_table <= 0) return AMT_E_INVALID;
    cqt_coef_kernel<<<(n_table * 32 + 255) / 256, 256, 0, (hipStream_t)stream>>>(phase_inc, length, n_table, coef);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_cqt_slices(const amt_cqt_args *args, void *stream) {
    if (!args || !args->wave || !args->src_frame || !args->phase_inc || !args->length || !args->out)
        return AMT_E_INVALID;
    const amt_cqt_args &q = *args;
    if (q.B <= 0 || q.L <= 0 || q.hop <= 0 || q.n_bins <= 0 || q.n_table <= 0) return AMT_E_INVALID;
    if (q.frames <= 0 || q.frames > AMT_CQT_MAXF) return AMT_E_UNSUPPORTED;
    if (q.wave_stride < (size_t)q.L) return AMT_E_SHAPE;
    CqtBlocksArgs a{};
    size_t lds;
    if (cqt_blocks_geometry(q.L, q.hop, &a.hshift, &a.blk_cap, &lds) != AMT_OK) {
        // hop not a power of two in 128..2048, or more hop-blocks than the LDS holds: the direct form
        cqt_slices_direct_kernel<<<dim3(q.n_bins, q.B), 256, 0, (hipStream_t)stream>>>(q);
        AMT_LAUNCH_CHECK();
        return AMT_OK;
    }
    if (!q.coef) return AMT_E_INVALID;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)cqt_blocks_kernel<true, false>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_set = true;
    }
    a.wave_stride = q.wave_stride; a.L = q.L; a.H = q.hop; a.T = 1 + q.L / q.hop;
    a.src_frame = q.src_frame; a.bin0 = q.bin0; a.ref = q.ref; a.out = q.out;
    a.frames = q.frames; a.n_bins = q.n_bins; a.n_table = q.n_table;
    cqt_blocks_kernel<true, false><<<dim3(q.n_bins, q.B), 256, lds, (hipStream_t)stream>>>(q.wave, q.phase_inc, q.length,
                                                                                     q.coef, a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// bytes of HBM workspace amt_cqt_window_max needs for (L, hop, n_bins, B): 0 while the block sums fit the LDS
extern "C" size_t amt_cqt_window_max_workspace(int L, int hop, int n_bins, int B) {
    int hshift, blk_cap;
    size_t lds;
    if (L <= 0 || hop <= 0 || n_bins <= 0 || B <= 0) return 0;
    if (cqt_blocks_geometry(L, hop, &hshift, &blk_cap, &lds) == AMT_OK) return 0;
    return (size_t)n_bins * B * ((size_t)blk_cap * 12 * sizeof(float) + ((size_t)blk_cap + 1) * 6 * sizeof(double));
}

extern "C" int amt_cqt_window_max(const float *wave, int B, int L, size_t wave_stride, int hop,
                                  const uint32_t *phase_inc, const int32_t *length, const float *coef, int n_bins,
                                  float *out_max, void *workspace, size_t workspace_bytes, void *stream) {
    if (!wave || !phase_inc || !length || !coef || !out_max) return AMT_E_INVALID;
    if (B <= 0 || L <= 0 || n_bins <= 0) return AMT_E_INVALID;
    if (wave_stride < (size_t)L) return AMT_E_SHAPE;
    if (!amt_is_pow2(hop) || hop < 128 || hop > 2048) return AMT_E_UNSUPPORTED;
    CqtBlocksArgs a{};
    size_t lds;
    const bool in_lds = cqt_blocks_geometry(L, hop, &a.hshift, &a.blk_cap, &lds) == AMT_OK;
    hipStream_t st = (hipStream_t)stream;
    AMT_HIP_CHECK(hipMemsetAsync(out_max, 0, (size_t)B * sizeof(float), st));
    a.wave_stride = wave_stride; a.L = L; a.H = hop; a.T = 1 + L / hop;
    a.out_max = (unsigned int *)out_max;
    if (in_lds) {
        static bool attr_set = false;
        if (!attr_set) {
            AMT_HIP_CHECK(hipFuncSetAttribute((const void *)cqt_blocks_kernel<false, false>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            attr_set = true;
        }
        cqt_blocks_kernel<false, false><<<dim3(n_bins, B), 256, lds, st>>>(wave, phase_inc, length, coef, a);
    } else {
        // a signal of more hop-blocks than the LDS holds (a whole song): block sums in the caller's HBM workspace
        const size_t need = amt_cqt_window_max_workspace(L, hop, n_bins, B);
        if (!workspace || workspace_bytes < need) return AMT_E_NOMEM;
        a.pf_ws = (double *)workspace;               // doubles first: 8-byte aligned whatever blk_cap is
        a.fg_ws = (float *)(a.pf_ws + (size_t)n_bins * B * ((size_t)a.blk_cap + 1) * 6);
        cqt_blocks_kernel<false, true><<<dim3(n_bins, B), 256, 4 * CM_STAGE * sizeof(float), st>>>(wave, phase_inc, length,
                                                                                                   coef, a);
    }
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
