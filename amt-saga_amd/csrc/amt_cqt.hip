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
#include <cstdlib>

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
    float *out_im;              // slices mode, complex form: imaginary parts (out then holds the real parts); null => magnitudes
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
            if (tid < a.frames) {
                a.out[((size_t)b * a.n_bins + k) * a.frames + tid] = 0.f;
                if (a.out_im) a.out_im[((size_t)b * a.n_bins + k) * a.frames + tid] = 0.f;
            }
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
    float fre = 0.f, fim = 0.f;                      // the last frame's complex sum (absolute sample phase), for the complex form
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
        fre = re; fim = im;
        return sqrtf(re * re + im * im) * scale;
    };
    if (SLICES) {
        if (tid < a.frames) {
            const int t = a.src_frame[b * a.frames + tid];
            float v = t >= 0 ? frame(t) : 0.f;
            const size_t oi = ((size_t)b * a.n_bins + k) * a.frames + tid;
            if (a.out_im) {
                // complex form: the sums above carry the phase of the ABSOLUTE sample index; the value handed out refers
                // to the frame's centre t H (the filter's phase is zero there, as for a centred filter bank): e^{+i phi t H},
                // integer-exact mod 2^32
                float re = 0.f, im = 0.f;
                if (t >= 0) {
                    const float turns = (float)((unsigned int)(t << hshift) * inc) * 2.3283064365386963e-10f;
                    const float c = __builtin_amdgcn_cosf(turns), sn = __builtin_amdgcn_sinf(turns);
                    re = (fre * c - fim * sn) * scale; im = (fre * sn + fim * c) * scale;
                    if (a.ref) { re = __fdiv_rn(re, a.ref[b]); im = __fdiv_rn(im, a.ref[b]); }
                }
                a.out[oi] = re; a.out_im[oi] = im;
            } else {
                if (a.ref) v = __fdiv_rn(v, a.ref[b]);
                a.out[oi] = v;
            }
        }
        return;
    }
    float vmax = 0.f;
    for (int t = tid; t < T; t += 256) vmax = fmaxf(vmax, frame(t));
    vmax = wave_max(vmax);
    if (lane == 0) atomicMax(a.out_max + b, __float_as_uint(vmax));   // non-negative floats order as their bits
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the whole-window maximum (amt_cqt_window_max_mfma): the block sums as ONE split-fp16 GEMM per window
// against a per-bin phasor table, on a BIN-INDEPENDENT block grid, so that the window is streamed once for a whole
// group of bins instead of once per bin (the VALU form above re-reads the 1 MB window for every bin: 4 B of L2->L1
// traffic and 3 packed FMAs per sample-bin; 157 ms per 1024 windows on the 1392-bin grid).
//
// Uniform blocks j = samples [jH, (j+1)H) whatever the bin.  With z0[m] = x[m] e^{-i phi m},
// z+-[m] = z0[m] e^{+-i theta (m + N_k/2)} and P(n) = sum_{m<n} z[m], frame t (samples [tH - N_k/2, tH - N_k/2 + N_k))
// is S_t = P(b_t) - P(a_t), and both ends sit at a FIXED offset inside their uniform block:
//     a_t = (t - hq) H + cut_s,   b_t = (t + eq) H + cut_e          (hq, eq, cut_s, cut_e depend on the bin only)
// so   S_t = PF(t + eq) - PF(t - hq) + Ge_{t+eq} - Gs_{t-hq},   PF(j) = sum_{j'<j} F_j'  (f64),
//     F_j = sum over block j,  Gs_j / Ge_j = sum over its first cut_s / cut_e samples.
// Per bin that is 18 real columns (F, Gs, Ge  x  z0, z+, z-  x  re, im) of a GEMM whose A operand is the window
// itself, [blocks][H], identical for every bin:  C[blocks][18 n_bins] = X[blocks][H] x Tab[H][18 n_bins], where
// Tab holds the phasors RELATIVE to the block start (the masks of Gs / Ge are zeros in the table) and the block's
// start phase (integer-exact, as above) is applied once per (bin, block) afterwards.
// Arithmetic: split-fp16 (x 2^s = h + l, both f16: 22 significant bits relative to the window's max |x|; the table
// likewise), three v_mfma_f32_16x16x32_f16 per product block (h h, h l, l h) into one f32 accumulator -- the dropped
// l l term and the flushed tiny l are < 2^-22 of max|x| max|tab| per product.
// Workgroup = (group of CQM_BINS bins, window), 512 threads: wave w owns M-tiles w, w+8, ... (its rows of the A slab
// live in a wave-private LDS region: no workgroup barrier for A), up to two leftover M-tiles are split over the waves
// by N-tile; the table slab of a k-step (32 samples) is shared and double-buffered: one barrier per k-step.  The
// next k-step's samples and table slab are in flight (registers) under the current k-step's MFMAs.
// ---------------------------------------------------------------------------------------------
#define CQM_BINS 7                      // bins per workgroup
#define CQM_NT 8                        // N-tiles of 16 columns: 7 x 18 = 126 of 128 columns used
#define CQM_MT 4                        // private M-tiles per wave
#define CQM_RS 2                        // shared (leftover) M-tiles
#define CQM_TSCALE 8.0f                 // table entries are scaled by 2^3 (|entry| <= 8)
typedef _Float16 cq_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cq_h4 __attribute__((ext_vector_type(4)));
typedef float cq_f4 __attribute__((ext_vector_type(4)));

#ifdef AMT_CQM_DIAG
#define CQM_DBG(a) ((a).debug)
#else
#define CQM_DBG(a) 0
#endif
struct CqmArgs {
    const float *wave; size_t wave_stride;
    const unsigned int *phase_inc; const int *length;
    const _Float16 *table;              // [groups][KS][2 planes][128 cols][32]
    const float *amax;                  // [B] max |x| of each window
    unsigned int *out_max;              // [B] float bits, zeroed by the caller
    int L, H, hshift, T, n_bins, nblk, q, r, KS;
    int debug;                          // diagnostic builds only (-DAMT_CQM_DIAG + env AMT_CQM_DEBUG: 1 no epilogue, 2 no MFMAs,
                                        // 4 no commit -- a timing breakdown with WRONG results); the product build ignores it
};

// table: column col of group g = bin g * CQM_BINS + col / 18, component col % 18 = set * 6 + kind * 2 + part
//   set 0: F (all samples), 1: Gs (samples < cut_s), 2: Ge (samples < cut_e);  kind 0: e^{-i phi i}, 1: x e^{+i theta i},
//   2: x e^{-i theta i};  part 0 / 1: real / imaginary
__global__ void cqt_mfma_table_kernel(const unsigned int *__restrict__ phase_inc, const int *__restrict__ length, int n_bins,
                                      int H, int hshift, _Float16 *__restrict__ table, size_t n_total) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_total) return;
    const int i = (int)(e & 31), col = (int)((e >> 5) & 127);
    const int KS = H >> 5;
    const size_t gc = e >> 12;                              // g * KS + c
    const int c = (int)(gc % KS), g = (int)(gc / KS);
    const int k = g * CQM_BINS + col / 18, comp = col % 18;
    float v = 0.f;
    if (col < CQM_BINS * 18 && k < n_bins) {
        const int nk = length[k], half = nk >> 1;
        const int hq = (half + H - 1) >> hshift, cut_s = (hq << hshift) - half;
        const int rem = nk - half, cut_e = rem - ((rem >> hshift) << hshift);
        const int ii = 32 * c + i, set = comp / 6, kind = (comp % 6) >> 1, part = comp & 1;
        const bool on = set == 0 || (set == 1 ? ii < cut_s : ii < cut_e);
        if (on) {
            const double ph = (double)phase_inc[k] * 4.656612873077393e-10 * (double)ii;     // phi ii in half-turns
            const double th = 2.0 / (double)nk * (double)ii;                                // theta ii
            double sn, cs;
            sincospi(kind == 0 ? -ph : (kind == 1 ? th - ph : -th - ph), &sn, &cs);
            v = (float)((part ? sn : cs) * (double)CQM_TSCALE);
        }
    }
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t base = gc * (size_t)(2 * 128 * 32) + (size_t)col * 32 + i;
    table[base] = h;
    table[base + 128 * 32] = l;
}

__global__ __launch_bounds__(256) void cqt_amax_kernel(const float *__restrict__ wave, size_t wave_stride, int L,
                                                        float *__restrict__ amax) {
    __shared__ float red[16];
    const float *x = wave + (size_t)blockIdx.x * wave_stride;
    float m = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = block_max(m, red);
    if (threadIdx.x == 0) amax[blockIdx.x] = m;
}

// f64 inclusive scan step inside 16-lane rows by DPP (row_shr: lanes without a source add 0): VALU-only, where
// __shfl_up goes through the LDS crossbar (ds_bpermute, ~100 cycles a step)
template <int N>
__device__ __forceinline__ double cqm_row_shr_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, false);
    return v + __hiloint2double(shi, slo);
}

// Q = private M-tiles per wave (0..CQM_MT), R = leftover M-tiles (0..CQM_RS): compile-time, so that the staging and
// MFMA loops are straight-line code (as run-time bounds the per-tile guards turned into branches around every load,
// each with its own s_waitcnt vmcnt(0): the prefetch serialised into eight L2 round trips per k-step -- 91 ms per
// 1024 windows on the 1392-bin grid against 65 for the same kernel with the guards compiled away).
template <int Q, int R>
__global__ __launch_bounds__(512, 2) void cqt_max_mfma_kernel(CqmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cqm_smem[];
    _Float16 *a_priv = (_Float16 *)cqm_smem;                       // [8 waves][2 planes][64 rows][32]          64 KB
    _Float16 *a_shr = a_priv + 8 * 2 * 64 * 32;                    // [2 buffers][2 planes][32 rows][32]         8 KB
    _Float16 *b_lds = a_shr + 2 * 2 * 32 * 32;                     // [2 buffers][2 planes][128 cols][32]       32 KB
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int g = blockIdx.x, b = blockIdx.y;
    const int L = a.L, H = a.H, hshift = a.hshift, KS = a.KS, nblk = a.nblk;
    constexpr int q = Q;
    const float *x = a.wave + (size_t)b * a.wave_stride;
    // per-window operand scale: max |x| 2^sx in [2^11, 2^12)
    int sx = 0;
    {
        const float am = a.amax[b];
        int e = 0;
        if (am > 0.f && am < INFINITY) { (void)frexpf(am, &e); sx = 12 - e; }
        sx = min(max(sx, -100), 100);
    }
    const float xs = ldexpf(1.0f, sx);

    typedef float cqm_f4u __attribute__((ext_vector_type(4), aligned(4)));
    // Staging of four samples of block `row` at k-step c in two halves: addr4() / the raw 16-byte load (issued a whole
    // k-step ahead, nothing but the load: any arithmetic on the loaded value at issue time makes the compiler wait for
    // it there) and fix4() at commit time (zeros past the end of the signal / the last block; a group that straddles L
    // -- only when L is not a multiple of 4 -- is shifted into place by selects).
    const bool ragged = (L & 3) != 0;                // uniform
    auto addr4 = [&](int row, int col4, int c) -> const cqm_f4u * {
        const int m = (row << hshift) + 32 * c + 4 * col4;
        const bool in = row < nblk && m < L;
        const unsigned int off = in ? (unsigned int)min(m, L - 4) * 4u : 0u;
        return (const cqm_f4u *)((const unsigned char *)x + off);
    };
    auto fix4 = [&](cq_f4 v, int row, int col4, int c) -> cq_f4 {
        const int m = (row << hshift) + 32 * c + 4 * col4;
        const bool in = row < nblk && m < L;
        if (ragged) {
            const int sh = m - min(m, L - 4);        // 0..3 when `in`
            const cq_f4 w = v;
            v.x = sh <= 0 ? w.x : (sh == 1 ? w.y : (sh == 2 ? w.z : w.w));
            v.y = sh <= 0 ? w.y : (sh == 1 ? w.z : (sh == 2 ? w.w : 0.f));
            v.z = sh <= 0 ? w.z : (sh == 1 ? w.w : 0.f);
            v.w = sh <= 0 ? w.w : 0.f;
        }
        const cq_f4 z = {0.f, 0.f, 0.f, 0.f};
        return in ? v : z;
    };
    // x 2^sx = h + l with h = f16(v) rounded toward zero (v_cvt_pkrtz_f16_f32 converts and packs two values per
    // instruction), l = f16(v - h): the residual of a truncation is below one ulp of h and l's 11 bits carry it to
    // 2^-21 of |v| or better
    typedef __fp16 cq_hp2 __attribute__((ext_vector_type(2)));
    struct cq_pair4 { cq_hp2 a, b; };
    auto split_store = [&](cq_f4 v, _Float16 *ph, _Float16 *pl) {
        v *= xs;
        const cq_hp2 h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
        const cq_hp2 l01 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01.x, v.y - (float)h01.y);
        const cq_hp2 l23 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23.x, v.w - (float)h23.y);
        *(cq_pair4 *)ph = cq_pair4{h01, h23};
        *(cq_pair4 *)pl = cq_pair4{l01, l23};
    };
    cq_f4 apre[2 * (Q > 0 ? Q : 1)], spre = {0.f, 0.f, 0.f, 0.f};
    const unsigned char *tabb = (const unsigned char *)(a.table + (size_t)g * KS * (2 * 128 * 32));
    const int srow = 16 * (8 * q + (tid >> 7)) + ((tid & 127) >> 3);          // this thread's row of the leftover tiles
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const int row0 = 16 * (wid + 8 * i) + (lane >> 3);
            apre[2 * i] = *addr4(row0, lane & 7, c);
            apre[2 * i + 1] = *addr4(row0 + 8, lane & 7, c);
        }
        if (R > 0 && tid < 128 * R) spre = *addr4(srow, tid & 7, c);
        // the table slab of the k-step (16 KB, a plain copy) goes global -> LDS directly (LDS-DMA: no registers; a wave
        // instruction lands 64 x 16 contiguous bytes), into the buffer the PREVIOUS k-step's MFMAs no longer read
        const unsigned char *src = tabb + (size_t)c * (2 * 128 * 32 * 2);
        unsigned char *dst = (unsigned char *)(b_lds + (size_t)(c & 1) * (2 * 128 * 32));
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (wid * 2 + u) * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + (wid * 2 + u) * 1024), 16, 0, 0);
    };
    auto commit = [&](int c) {                       // staged registers -> LDS (wave-private A; leftover tiles' A into buffer c & 1)
        _Float16 *ap = a_priv + (size_t)wid * (2 * 64 * 32);
#pragma unroll
        for (int i = 0; i < Q; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rl = 16 * i + 8 * u + (lane >> 3);
                const cq_f4 v = fix4(apre[2 * i + u], 16 * (wid + 8 * i) + 8 * u + (lane >> 3), lane & 7, c);
                split_store(v, ap + rl * 32 + 4 * (lane & 7), ap + (64 + rl) * 32 + 4 * (lane & 7));
            }
        }
        if (R > 0 && tid < 128 * R) {
            _Float16 *sp = a_shr + (size_t)(c & 1) * (2 * 32 * 32);
            const int rl = 16 * (tid >> 7) + ((tid & 127) >> 3);
            split_store(fix4(spre, srow, tid & 7, c), sp + rl * 32 + 4 * (tid & 7), sp + (32 + rl) * 32 + 4 * (tid & 7));
        }
    };

    cq_f4 acc[Q > 0 ? Q : 1][CQM_NT], accs[R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < (Q > 0 ? Q : 1); ++i)
#pragma unroll
        for (int n = 0; n < CQM_NT; ++n) acc[i][n] = cq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < (R > 0 ? R : 1); ++s_) accs[s_] = cq_f4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    const int fr = (lane & 15) * 32 + 8 * (lane >> 4);               // fragment offset inside a [16][32] tile (halfs)
    for (int c = 0; c < KS; ++c) {
        if (!(CQM_DBG(a) & 4)) commit(c);
        __syncthreads();
        if (c + 1 < KS) fetch(c + 1);
        if (CQM_DBG(a) & 2) continue;
        const _Float16 *ap = a_priv + (size_t)wid * (2 * 64 * 32);
        const _Float16 *sp = a_shr + (size_t)(c & 1) * (2 * 32 * 32);
        const _Float16 *bp = b_lds + (size_t)(c & 1) * (2 * 128 * 32);
        cq_h8 ah[Q > 0 ? Q : 1], al[Q > 0 ? Q : 1];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            ah[i] = *(const cq_h8 *)(ap + 16 * i * 32 + fr);
            al[i] = *(const cq_h8 *)(ap + (64 + 16 * i) * 32 + fr);
        }
#pragma unroll
        for (int n = 0; n < CQM_NT; ++n) {
            if (Q == 0) break;
            const cq_h8 bh = *(const cq_h8 *)(bp + 16 * n * 32 + fr);
            const cq_h8 bl = *(const cq_h8 *)(bp + (128 + 16 * n) * 32 + fr);
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][n], 0, 0, 0);
            }
        }
        // leftover M-tiles: this wave takes N-tile `wid` of each
        if (R > 0) {
            const cq_h8 bh = *(const cq_h8 *)(bp + 16 * wid * 32 + fr);
            const cq_h8 bl = *(const cq_h8 *)(bp + (128 + 16 * wid) * 32 + fr);
#pragma unroll
            for (int s_ = 0; s_ < R; ++s_) {
                const cq_h8 sh = *(const cq_h8 *)(sp + 16 * s_ * 32 + fr);
                const cq_h8 sl = *(const cq_h8 *)(sp + (32 + 16 * s_) * 32 + fr);
                accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sh, bh, accs[s_], 0, 0, 0);
                accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sh, bl, accs[s_], 0, 0, 0);
                accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sl, bh, accs[s_], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                 // every wave is done with the operand buffers: reuse them below
    if (CQM_DBG(a) & 1) {                            // timing breakdown only
        float v = 0.f;
        for (int i = 0; i < (Q > 0 ? Q : 1); ++i) for (int n = 0; n < CQM_NT; ++n) v += acc[i][n].x;
        if (v == 12345.f) atomicMax(a.out_max + b, 1u);
        return;
    }

    // ---- epilogue, bin by bin: raw block sums -> LDS, f64 prefix of the phase-corrected F, frames -----------------------
    // Two sets of (raw sums Rb [nrow][18], prefix pf [nblk + 1][6] f64), alternating between bins: while three waves
    // scan bin bb, a fourth evaluates the frames of bin bb - 1 that did not fit its 512-thread round (T = 516: four
    // frames would otherwise cost every wave a second round), and the dump of bin bb + 1 needs no barrier of its own.
    constexpr int nrow = 16 * (8 * Q + R);
    constexpr size_t RB_BYTES = ((size_t)nrow * 18 * 4 + 15) & ~(size_t)15;
    const size_t PF_BYTES = (((size_t)nblk + 1) * 6 * sizeof(double) + 15) & ~(size_t)15;
    const float unscale = ldexpf(1.0f, -sx) / CQM_TSCALE;
    float vmax = 0.f;
    int nk = 1, half = 0, hq = 0, eq = 0;            // of the bin whose scan / frames run now
    unsigned int inc = 0;
    float inv_nk = 1.f;
    // v mod nk for 0 <= v < 2^24 (block starts + N_k/2 stay far below): quotient from the float reciprocal, corrected
    // by one step either way -- a dozen instructions against ~40 for the integer division
    auto umod = [&](unsigned int v) -> unsigned int {
        int qv = (int)((float)v * inv_nk);
        int rv = (int)v - qv * nk;
        rv += rv < 0 ? nk : 0;
        rv -= rv >= nk ? nk : 0;
        return (unsigned int)rv;
    };
    // start phases of block j (integer-exact): osc = e^{-i phi jH}, wp = e^{+i theta (jH + N_k/2)}
    auto anchors = [&](int j, amt_v2 &osc, amt_v2 &wp) {
        const unsigned int ms = (unsigned int)j << hshift;
        const float turns = (float)(ms * inc) * 2.3283064365386963e-10f;
        osc = amt_v2{__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns)};
        const float wt = (float)umod(ms + (unsigned int)half) * inv_nk;
        wp = amt_v2{__builtin_amdgcn_cosf(wt), __builtin_amdgcn_sinf(wt)};
    };
    auto set_bin = [&](int k) {
        nk = a.length[k]; inc = a.phase_inc[k];
        half = nk >> 1; hq = (half + H - 1) >> hshift; eq = (nk - half) >> hshift;
        inv_nk = 1.0f / (float)nk;
    };
    auto frame = [&](int t, const float *Rb, const double *pf) -> float {
        const int js = t - hq, je = t + eq;
        const int i0 = min(max(js, 0), nblk), i1 = min(max(je, 0), nblk);
        float sfr[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) sfr[e] = (float)(pf[(size_t)i1 * 6 + e] - pf[(size_t)i0 * 6 + e]);
        // the heads of the end block (+) and of the start block (-): raw sums x their block's start phases
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int j = side ? js : je;
            if (j < 0 || j >= nblk) continue;
            amt_v2 osc, wp;
            anchors(j, osc, wp);
            const amt_v2 op = cm_mul(osc, wp), om = cm_mul(osc, amt_v2{wp.x, -wp.y});
            const float *p = Rb + j * 18 + (side ? 6 : 12);
            const float sg = side ? -unscale : unscale;
            const amt_v2 g0 = cm_mul(amt_v2{p[0], p[1]}, osc) * sg;
            const amt_v2 g1 = cm_mul(amt_v2{p[2], p[3]}, op) * sg;
            const amt_v2 g2 = cm_mul(amt_v2{p[4], p[5]}, om) * sg;
            sfr[0] += g0.x; sfr[1] += g0.y; sfr[2] += g1.x; sfr[3] += g1.y; sfr[4] += g2.x; sfr[5] += g2.y;
        }
        const float dt = (float)umod((unsigned int)t << hshift) * inv_nk;
        const float sd = __builtin_amdgcn_sinf(dt), cd = __builtin_amdgcn_cosf(dt);
        const float ar = cd * sfr[2] + sd * sfr[3], ai = cd * sfr[3] - sd * sfr[2];
        const float br = cd * sfr[4] - sd * sfr[5], bi = cd * sfr[5] + sd * sfr[4];
        const float re = 0.5f * sfr[0] - 0.25f * (ar + br);
        const float im = 0.5f * sfr[1] - 0.25f * (ai + bi);
        return sqrtf(re * re + im * im) * (2.0f / sqrtf((float)nk));
    };
    const int nb_here = min(CQM_BINS, a.n_bins - g * CQM_BINS);      // bins of this group (>= 1)
    for (int bb = 0; bb <= nb_here; ++bb) {
        float *Rb = (float *)(cqm_smem + (size_t)(bb & 1) * (RB_BYTES + PF_BYTES));
        double *pf = (double *)((unsigned char *)Rb + RB_BYTES);
        const float *Rp = (const float *)(cqm_smem + (size_t)((bb ^ 1) & 1) * (RB_BYTES + PF_BYTES));     // previous bin's set
        const double *pfp = (const double *)((const unsigned char *)Rp + RB_BYTES);
        if (bb < nb_here) {                          // dump bin bb's 18 columns (no barrier needed first: the set was last
            const int c0 = 18 * bb, nt0 = c0 >> 4;   // read two barriers ago)
#pragma unroll
            for (int n = 0; n < CQM_NT; ++n) {
                if (n != nt0 && n != nt0 + 1) continue;  // uniform
                const int cc = 16 * n + (lane & 15) - c0;
                if (cc < 0 || cc >= 18) continue;
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const int row = 16 * (wid + 8 * i) + 4 * (lane >> 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Rb[(row + e) * 18 + cc] = acc[i][n][e];
                }
            }
            if (R > 0 && (wid == nt0 || wid == nt0 + 1)) {
                const int cc = 16 * wid + (lane & 15) - c0;
                if (cc >= 0 && cc < 18) {
#pragma unroll
                    for (int s_ = 0; s_ < R; ++s_) {
                        const int row = 16 * (8 * q + s_) + 4 * (lane >> 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) Rb[(row + e) * 18 + cc] = accs[s_][e];
                    }
                }
            }
        }
        __syncthreads();
        if (bb < nb_here && wid < 3) {
            // PF[i] = sum_{j < i} F_j in f64, one wave per complex component (z0, z+, z-): a lane owns PER consecutive
            // blocks, applies their start phases (exact at its first block, a short recurrence after it), sums them;
            // the 64 lane totals are scanned (DPP inside rows of 16, shuffles across), the lane writes its prefixes
            set_bin(g * CQM_BINS + bb);
            constexpr int PER = (nrow + 63) / 64;
            const int j0 = lane * PER;
            amt_v2 osc, wp, osc1, wp1;
            anchors(min(j0, nblk), osc, wp);
            anchors(1, osc1, wp1);                   // osc1 = e^{-i phi H}
            const float wts = (float)umod((unsigned int)H) * inv_nk;
            const amt_v2 wstep = amt_v2{__builtin_amdgcn_cosf(wts), __builtin_amdgcn_sinf(wts)};     // e^{+i theta H}, integer-exact
            amt_v2 an = wid == 0 ? osc : (wid == 1 ? cm_mul(osc, wp) : cm_mul(osc, amt_v2{wp.x, -wp.y}));
            const amt_v2 st = wid == 0 ? osc1 : (wid == 1 ? cm_mul(osc1, wstep) : cm_mul(osc1, amt_v2{wstep.x, -wstep.y}));
            amt_v2 z[PER];
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = j0 + i;
                const int jc = min(j, nrow - 1);
                const amt_v2 raw = amt_v2{Rb[jc * 18 + 2 * wid], Rb[jc * 18 + 2 * wid + 1]};
                z[i] = j < nblk ? cm_mul(raw, an) * unscale : amt_v2{0.f, 0.f};
                sr += (double)z[i].x; si += (double)z[i].y;
                an = cm_mul(an, st);
            }
            double vr = sr, vi = si;
            vr = cqm_row_shr_add<1>(vr); vi = cqm_row_shr_add<1>(vi);
            vr = cqm_row_shr_add<2>(vr); vi = cqm_row_shr_add<2>(vi);
            vr = cqm_row_shr_add<4>(vr); vi = cqm_row_shr_add<4>(vi);
            vr = cqm_row_shr_add<8>(vr); vi = cqm_row_shr_add<8>(vi);
            {   // totals of the rows before this lane's row: lane 15 / 31 / 47 hold the row sums
                const double r0 = __shfl(vr, 15, 64), r1 = __shfl(vr, 31, 64), r2 = __shfl(vr, 47, 64);
                const double i0_ = __shfl(vi, 15, 64), i1_ = __shfl(vi, 31, 64), i2_ = __shfl(vi, 47, 64);
                const int row = lane >> 4;
                vr += (row > 0 ? r0 : 0.0) + (row > 1 ? r1 : 0.0) + (row > 2 ? r2 : 0.0);
                vi += (row > 0 ? i0_ : 0.0) + (row > 1 ? i1_ : 0.0) + (row > 2 ? i2_ : 0.0);
            }
            double rr = vr - sr, ri = vi - si;       // blocks before this lane's run
            if (lane == 0) { pf[2 * wid] = 0.0; pf[2 * wid + 1] = 0.0; }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = j0 + i;
                rr += (double)z[i].x; ri += (double)z[i].y;
                if (j < nblk) { pf[(size_t)(j + 1) * 6 + 2 * wid] = rr; pf[(size_t)(j + 1) * 6 + 2 * wid + 1] = ri; }
            }
        } else if (bb > 0 && wid == 3) {
            // frames 512 .. T-1 of the previous bin, from its (still intact) set
            set_bin(g * CQM_BINS + bb - 1);
            for (int t = 512 + lane; t < a.T; t += 64) vmax = fmaxf(vmax, frame(t, Rp, pfp));
        }
        __syncthreads();
        if (bb < nb_here) {
            set_bin(g * CQM_BINS + bb);
            if (tid < a.T) vmax = fmaxf(vmax, frame(tid, Rb, pf));
        }
    }
    vmax = wave_max(vmax);
    if (lane == 0) atomicMax(a.out_max + b, __float_as_uint(vmax));
}

// Fallback for signals whose block sums do not fit the LDS (a whole song handed to slice_C): every requested
// frame summed directly over its own N_k samples, one workgroup per (bin, window).  Any L; 8 N_k sample visits.
__global__ __launch_bounds__(256) void cqt_slices_direct_kernel(amt_cqt_args a, float *out_im) {
    __shared__ float red[4][2];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int kt = (a.bin0 ? a.bin0[b] : 0) + k;
    float *o = a.out + ((size_t)b * a.n_bins + k) * a.frames;
    if (kt < 0 || kt >= a.n_table) {                // uniform
        if (tid < a.frames) { o[tid] = 0.f; if (out_im) out_im[((size_t)b * a.n_bins + k) * a.frames + tid] = 0.f; }
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
            if (out_im) {                                  // complex form, phase referred to the frame centre t hop
                float cr = 0.f, ci = 0.f;
                if (t >= 0) {
                    const float turns = (float)((unsigned int)((long long)t * a.hop) * inc) * 2.3283064365386963e-10f;
                    const float c = __builtin_amdgcn_cosf(turns), sn = __builtin_amdgcn_sinf(turns);
                    cr = (r * c - i * sn) * scale; ci = (r * sn + i * c) * scale;
                    if (a.ref) { cr = __fdiv_rn(cr, a.ref[b]); ci = __fdiv_rn(ci, a.ref[b]); }
                }
                o[j] = cr;
                out_im[((size_t)b * a.n_bins + k) * a.frames + j] = ci;
                continue;
            }
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

static int cqt_slices_impl(const amt_cqt_args *args, float *out_im, void *stream);
extern "C" int amt_cqt_slices(const amt_cqt_args *args, void *stream) { return cqt_slices_impl(args, nullptr, stream); }
extern "C" int amt_cqt_slices_complex(const amt_cqt_args *args, float *out_im, void *stream) {
    if (!out_im) return AMT_E_INVALID;
    return cqt_slices_impl(args, out_im, stream);
}
static int cqt_slices_impl(const amt_cqt_args *args, float *out_im, void *stream) {
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
        cqt_slices_direct_kernel<<<dim3(q.n_bins, q.B), 256, 0, (hipStream_t)stream>>>(q, out_im);
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
    a.src_frame = q.src_frame; a.bin0 = q.bin0; a.ref = q.ref; a.out = q.out; a.out_im = out_im;
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

// ---- MFMA form: geometry, table, launch ---------------------------------------------------------------------------
static int cqm_geometry(int L, int hop, int *hshift, int *nblk, int *q, int *r) {
    if (!amt_is_pow2(hop) || hop < 256 || hop > 2048) return AMT_E_UNSUPPORTED;
    *hshift = 0;
    while ((1 << *hshift) < hop) ++*hshift;
    *nblk = (L + hop - 1) / hop;
    const int n_mt = (*nblk + 15) / 16;
    *q = n_mt / 8; *r = n_mt % 8;
    if (*r > CQM_RS) { *q += 1; *r = 0; }
    if (*q > CQM_MT) return AMT_E_UNSUPPORTED;                 // more blocks than the wave-private slabs hold: VALU form
    return AMT_OK;
}

extern "C" size_t amt_cqt_mfma_table_bytes(int hop, int n_bins) {
    if (!amt_is_pow2(hop) || hop < 256 || hop > 2048 || n_bins <= 0) return 0;
    const size_t groups = (size_t)(n_bins + CQM_BINS - 1) / CQM_BINS;
    return groups * (size_t)(hop >> 5) * (2 * 128 * 32) * sizeof(_Float16);
}

extern "C" int amt_cqt_mfma_table(const uint32_t *phase_inc, const int32_t *length, int n_bins, int hop, void *table,
                                  void *stream) {
    if (!phase_inc || !length || !table || n_bins <= 0) return AMT_E_INVALID;
    const size_t bytes = amt_cqt_mfma_table_bytes(hop, n_bins);
    if (!bytes) return AMT_E_UNSUPPORTED;
    int hshift = 0;
    while ((1 << hshift) < hop) ++hshift;
    const size_t n_total = bytes / sizeof(_Float16) / 2;       // one thread per (h, l) pair
    cqt_mfma_table_kernel<<<(unsigned)((n_total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        phase_inc, length, n_bins, hop, hshift, (_Float16 *)table, n_total);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

extern "C" int amt_cqt_window_max_mfma(const float *wave, int B, int L, size_t wave_stride, int hop,
                                       const uint32_t *phase_inc, const int32_t *length, const void *table, int n_bins,
                                       float *out_max, float *amax_scratch, void *stream) {
    if (!wave || !phase_inc || !length || !table || !out_max || !amax_scratch) return AMT_E_INVALID;
    if (B <= 0 || L <= 0 || n_bins <= 0) return AMT_E_INVALID;
    if (wave_stride < (size_t)L) return AMT_E_SHAPE;
    CqmArgs a{};
    if (cqm_geometry(L, hop, &a.hshift, &a.nblk, &a.q, &a.r) != AMT_OK) return AMT_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_ops = (size_t)(8 * 2 * 64 * 32 + 2 * 2 * 32 * 32 + 2 * 2 * 128 * 32) * sizeof(_Float16);
    const size_t nrow = 16 * (size_t)(8 * a.q + a.r);
    const size_t lds_epi = 2 * (((nrow * 18 * 4 + 15) & ~(size_t)15) + ((((size_t)a.nblk + 1) * 6 * sizeof(double) + 15) & ~(size_t)15));
    const size_t lds = lds_ops > lds_epi ? lds_ops : lds_epi;
    void (*kern)(CqmArgs) = nullptr;
    switch (a.q * 3 + a.r) {
#define CQM_CASE(Q_, R_) case Q_ * 3 + R_: kern = cqt_max_mfma_kernel<Q_, R_>; break;
        CQM_CASE(0, 1) CQM_CASE(0, 2)
        CQM_CASE(1, 0) CQM_CASE(1, 1) CQM_CASE(1, 2) CQM_CASE(2, 0) CQM_CASE(2, 1) CQM_CASE(2, 2)
        CQM_CASE(3, 0) CQM_CASE(3, 1) CQM_CASE(3, 2) CQM_CASE(4, 0) CQM_CASE(4, 1) CQM_CASE(4, 2)
#undef CQM_CASE
        default: return AMT_E_UNSUPPORTED;
    }
    AMT_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    AMT_HIP_CHECK(hipMemsetAsync(out_max, 0, (size_t)B * sizeof(float), st));
    cqt_amax_kernel<<<B, 256, 0, st>>>(wave, wave_stride, L, amax_scratch);
    a.wave = wave; a.wave_stride = wave_stride; a.phase_inc = phase_inc; a.length = length;
    a.table = (const _Float16 *)table; a.amax = amax_scratch; a.out_max = (unsigned int *)out_max;
    a.L = L; a.H = hop; a.T = 1 + L / hop; a.n_bins = n_bins; a.KS = hop >> 5;
    a.debug = 0;
#ifdef AMT_CQM_DIAG
    { const char *e = getenv("AMT_CQM_DEBUG"); a.debug = e ? atoi(e) : 0; }
#endif
    const int groups = (n_bins + CQM_BINS - 1) / CQM_BINS;
    kern<<<dim3(groups, B), 512, lds, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
