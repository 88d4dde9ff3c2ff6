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

struct CqmArgs {
    const float *wave; size_t wave_stride;
    const unsigned int *phase_inc; const int *length;
    const _Float16 *table;              // [groups][KS][2 planes][128 cols][32]
    const float *amax;                  // [B] max |x| of each window
    unsigned int *out_max;              // [B] float bits, zeroed by the caller
    int L, H, hshift, T, n_bins, nblk, q, r, KS;
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

__global__ __launch_bounds__(512, 2) void cqt_max_mfma_kernel(CqmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cqm_smem[];
    _Float16 *a_priv = (_Float16 *)cqm_smem;                       // [8 waves][2 planes][64 rows][32]          64 KB
    _Float16 *a_shr = a_priv + 8 * 2 * 64 * 32;                    // [2 buffers][2 planes][32 rows][32]         8 KB
    _Float16 *b_lds = a_shr + 2 * 2 * 32 * 32;                     // [2 buffers][2 planes][128 cols][32]       32 KB
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int g = blockIdx.x, b = blockIdx.y;
    const int L = a.L, H = a.H, hshift = a.hshift, KS = a.KS, q = a.q, r = a.r, nblk = a.nblk;
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
    auto load4 = [&](int row, int col4, int c) -> cq_f4 {
        const long m = (long)row * H + 32 * c + 4 * col4;
        if (row < nblk && m + 3 < L) return *(const cqm_f4u *)(x + m);
        cq_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nblk) {
            if (m < L) v.x = x[m];
            if (m + 1 < L) v.y = x[m + 1];
            if (m + 2 < L) v.z = x[m + 2];
        }
        return v;
    };
    auto split_store = [&](cq_f4 v, _Float16 *ph, _Float16 *pl) {
        v *= xs;
        cq_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        cq_h4 l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z),
                   (_Float16)(v.w - (float)h.w)};
        *(cq_h4 *)ph = h;
        *(cq_h4 *)pl = l;
    };
    cq_f4 apre[2 * CQM_MT], spre;
    uint4 bpre[2];
    const uint4 *tab4 = (const uint4 *)(a.table + (size_t)g * KS * (2 * 128 * 32));
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < CQM_MT; ++i)
            if (i < q) {
                const int row0 = 16 * (wid + 8 * i) + (lane >> 3);
                apre[2 * i] = load4(row0, lane & 7, c);
                apre[2 * i + 1] = load4(row0 + 8, lane & 7, c);
            }
        if (tid < 128 * r) spre = load4(16 * (8 * q + (tid >> 7)) + ((tid & 127) >> 3), tid & 7, c);
        const uint4 *src = tab4 + (size_t)c * (2 * 128 * 32 * 2 / 16);
        bpre[0] = src[tid];
        bpre[1] = src[tid + 512];
    };
    auto commit = [&](int c) {                       // registers -> LDS (wave-private A; shared A / B into buffer c & 1)
        _Float16 *ap = a_priv + (size_t)wid * (2 * 64 * 32);
#pragma unroll
        for (int i = 0; i < CQM_MT; ++i)
            if (i < q) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int rl = 16 * i + 8 * u + (lane >> 3);
                    split_store(apre[2 * i + u], ap + rl * 32 + 4 * (lane & 7), ap + (64 + rl) * 32 + 4 * (lane & 7));
                }
            }
        if (tid < 128 * r) {
            _Float16 *sp = a_shr + (size_t)(c & 1) * (2 * 32 * 32);
            const int rl = 16 * (tid >> 7) + ((tid & 127) >> 3);
            split_store(spre, sp + rl * 32 + 4 * (tid & 7), sp + (32 + rl) * 32 + 4 * (tid & 7));
        }
        uint4 *bd = (uint4 *)(b_lds + (size_t)(c & 1) * (2 * 128 * 32));
        bd[tid] = bpre[0];
        bd[tid + 512] = bpre[1];
    };

    cq_f4 acc[CQM_MT][CQM_NT], accs[CQM_RS];
#pragma unroll
    for (int i = 0; i < CQM_MT; ++i)
#pragma unroll
        for (int n = 0; n < CQM_NT; ++n) acc[i][n] = cq_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < CQM_RS; ++s_) accs[s_] = cq_f4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    const int fr = (lane & 15) * 32 + 8 * (lane >> 4);               // fragment offset inside a [16][32] tile (halfs)
    for (int c = 0; c < KS; ++c) {
        commit(c);
        __syncthreads();
        if (c + 1 < KS) fetch(c + 1);
        const _Float16 *ap = a_priv + (size_t)wid * (2 * 64 * 32);
        const _Float16 *sp = a_shr + (size_t)(c & 1) * (2 * 32 * 32);
        const _Float16 *bp = b_lds + (size_t)(c & 1) * (2 * 128 * 32);
        cq_h8 ah[CQM_MT], al[CQM_MT];
#pragma unroll
        for (int i = 0; i < CQM_MT; ++i)
            if (i < q) {
                ah[i] = *(const cq_h8 *)(ap + 16 * i * 32 + fr);
                al[i] = *(const cq_h8 *)(ap + (64 + 16 * i) * 32 + fr);
            }
#pragma unroll
        for (int n = 0; n < CQM_NT; ++n) {
            const cq_h8 bh = *(const cq_h8 *)(bp + 16 * n * 32 + fr);
            const cq_h8 bl = *(const cq_h8 *)(bp + (128 + 16 * n) * 32 + fr);
#pragma unroll
            for (int i = 0; i < CQM_MT; ++i)
                if (i < q) {
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][n], 0, 0, 0);
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][n], 0, 0, 0);
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][n], 0, 0, 0);
                }
        }
        // leftover M-tiles: this wave takes N-tile `wid` of each
        {
            const cq_h8 bh = *(const cq_h8 *)(bp + 16 * wid * 32 + fr);
            const cq_h8 bl = *(const cq_h8 *)(bp + (128 + 16 * wid) * 32 + fr);
#pragma unroll
            for (int s_ = 0; s_ < CQM_RS; ++s_)
                if (s_ < r) {
                    const cq_h8 sh = *(const cq_h8 *)(sp + 16 * s_ * 32 + fr);
                    const cq_h8 sl = *(const cq_h8 *)(sp + (32 + 16 * s_) * 32 + fr);
                    accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sh, bh, accs[s_], 0, 0, 0);
                    accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sh, bl, accs[s_], 0, 0, 0);
                    accs[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sl, bh, accs[s_], 0, 0, 0);
                }
        }
    }
    __syncthreads();                                 // every wave is done with the operand buffers: reuse them below

    // ---- epilogue, bin by bin: raw block sums -> LDS, start phases, f64 prefix, frames ------------------------------
    const int nrow = 16 * (8 * q + r);
    float *R = (float *)cqm_smem;                    // [nrow][18]
    double *pf = (double *)(cqm_smem + (((size_t)nrow * 18 * 4 + 15) & ~(size_t)15));      // [nblk + 1][6]
    const float unscale = ldexpf(1.0f, -sx) / CQM_TSCALE;
    float vmax = 0.f;
    for (int bb = 0; bb < CQM_BINS; ++bb) {
        const int k = g * CQM_BINS + bb;
        if (k >= a.n_bins) break;                    // uniform
        const int c0 = 18 * bb, nt0 = c0 >> 4;
#pragma unroll
        for (int n = 0; n < CQM_NT; ++n) {
            if (n != nt0 && n != nt0 + 1) continue;  // uniform
            const int cc = 16 * n + (lane & 15) - c0;
            if (cc < 0 || cc >= 18) continue;
#pragma unroll
            for (int i = 0; i < CQM_MT; ++i)
                if (i < q) {
                    const int row = 16 * (wid + 8 * i) + 4 * (lane >> 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) R[(row + e) * 18 + cc] = acc[i][n][e];
                }
        }
        if (wid == nt0 || wid == nt0 + 1) {
            const int cc = 16 * wid + (lane & 15) - c0;
            if (cc >= 0 && cc < 18) {
#pragma unroll
                for (int s_ = 0; s_ < CQM_RS; ++s_)
                    if (s_ < r) {
                        const int row = 16 * (8 * q + s_) + 4 * (lane >> 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) R[(row + e) * 18 + cc] = accs[s_][e];
                    }
            }
        }
        __syncthreads();
        const int nk = a.length[k];
        const unsigned int inc = a.phase_inc[k];
        const int half = nk >> 1;
        const int hq = (half + H - 1) >> hshift;
        const int eq = (nk - half) >> hshift;
        const float inv_nk = 1.0f / (float)nk;
        // block start phases (integer-exact): e^{-i phi jH} and e^{+-i theta (jH + N_k/2)}
        for (int j = tid; j < nblk; j += 512) {
            const unsigned int ms = (unsigned int)j << hshift;
            const float turns = (float)(ms * inc) * 2.3283064365386963e-10f;
            const amt_v2 osc = amt_v2{__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns)};
            const float wt = (float)((int)(((long)ms + half) % nk)) * inv_nk;
            const amt_v2 wp = amt_v2{__builtin_amdgcn_cosf(wt), __builtin_amdgcn_sinf(wt)};
            const amt_v2 op = cm_mul(osc, wp), om = cm_mul(osc, amt_v2{wp.x, -wp.y});
            float *rj = R + j * 18;
#pragma unroll
            for (int set = 0; set < 3; ++set) {
                float *p = rj + 6 * set;
                const amt_v2 s0 = cm_mul(amt_v2{p[0], p[1]}, osc) * unscale;
                const amt_v2 s1 = cm_mul(amt_v2{p[2], p[3]}, op) * unscale;
                const amt_v2 s2 = cm_mul(amt_v2{p[4], p[5]}, om) * unscale;
                p[0] = s0.x; p[1] = s0.y; p[2] = s1.x; p[3] = s1.y; p[4] = s2.x; p[5] = s2.y;
            }
        }
        __syncthreads();
        if (tid < 192) {                             // PF[i] = sum_{j < i} F_j in f64 (as in cqt_blocks_kernel)
            const int e = tid >> 5, l = tid & 31;
            const int per = (nblk + 31) >> 5;
            const int i0 = min(l * per, nblk), i1 = min(i0 + per, nblk);
            double sum = 0.0;
            for (int i = i0; i < i1; ++i) sum += (double)R[i * 18 + e];
            double v = sum;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const double u = __shfl_up(v, off, 32);
                if (l >= off) v += u;
            }
            double run = v - sum;
            if (l == 0) pf[e] = 0.0;
            for (int i = i0; i < i1; ++i) {
                run += (double)R[i * 18 + e];
                pf[(size_t)(i + 1) * 6 + e] = run;
            }
        }
        __syncthreads();
        const float scale = 2.0f / sqrtf((float)nk);
        for (int t = tid; t < a.T; t += 512) {
            const int js = t - hq, je = t + eq;
            const int i0 = min(max(js, 0), nblk), i1 = min(max(je, 0), nblk);
            float sfr[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                float v = (float)(pf[(size_t)i1 * 6 + e] - pf[(size_t)i0 * 6 + e]);
                if (je >= 0 && je < nblk) v += R[je * 18 + 12 + e];
                if (js >= 0 && js < nblk) v -= R[js * 18 + 6 + e];
                sfr[e] = v;
            }
            const float dt = (float)((int)(((long)t << hshift) % nk)) * inv_nk;
            const float sd = __builtin_amdgcn_sinf(dt), cd = __builtin_amdgcn_cosf(dt);
            const float ar = cd * sfr[2] + sd * sfr[3], ai = cd * sfr[3] - sd * sfr[2];
            const float br = cd * sfr[4] - sd * sfr[5], bi = cd * sfr[5] + sd * sfr[4];
            const float re = 0.5f * sfr[0] - 0.25f * (ar + br);
            const float im = 0.5f * sfr[1] - 0.25f * (ai + bi);
            vmax = fmaxf(vmax, sqrtf(re * re + im * im) * scale);
        }
        __syncthreads();                             // R / pf are rewritten for the next bin
    }
    vmax = wave_max(vmax);
    if (lane == 0) atomicMax(a.out_max + b, __float_as_uint(vmax));
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
    const size_t lds_epi = ((nrow * 18 * 4 + 15) & ~(size_t)15) + ((size_t)a.nblk + 1) * 6 * sizeof(double);
    const size_t lds = lds_ops > lds_epi ? lds_ops : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)cqt_max_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          159 * 1024));
        attr_set = true;
    }
    AMT_HIP_CHECK(hipMemsetAsync(out_max, 0, (size_t)B * sizeof(float), st));
    cqt_amax_kernel<<<B, 256, 0, st>>>(wave, wave_stride, L, amax_scratch);
    a.wave = wave; a.wave_stride = wave_stride; a.phase_inc = phase_inc; a.length = length;
    a.table = (const _Float16 *)table; a.amax = amax_scratch; a.out_max = (unsigned int *)out_max;
    a.L = L; a.H = hop; a.T = 1 + L / hop; a.n_bins = n_bins; a.KS = hop >> 5;
    const int groups = (n_bins + CQM_BINS - 1) / CQM_BINS;
    cqt_max_mfma_kernel<<<dim3(groups, B), 512, lds, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
