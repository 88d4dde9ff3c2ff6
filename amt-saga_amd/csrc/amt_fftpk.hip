// Packed-image FFT-domain form of the timing head's 64 -> 64 (4 x 16) convolutions on its 10 x 64 images (conv mode 3).
//
// Replaces, for the eleven layers per timing net it is built for, Conv2D(64 -> 64, (4, 16), 'same') + BatchNormalization
// + sigmoid (+ Add + BatchNormalization) of /root/reference/RDCNN.py:186-198 on the 10 x 64 images that follow the first
// pooling of timing_classifier.py:13-36 -- in the direct split-fp16 form 0.91 ms per 1024 windows and layer at 0.46 of the
// matrix-pipe roof, 22 % of a C3 step.  scripts/fftconv_packed_model.py is the numpy model of the algebra:
//
//   * the WHOLE image is one sequence of NF = 1152 = 12 x 96 points: image row h occupies positions 96 h .. 96 h + 63,
//     the 32 positions behind every row and the row slots 10, 11 are zeros.  The 2-D 'same' convolution is then the 1-D
//     CIRCULAR convolution of that sequence with the kernel's taps at offsets (dy - 1) 96 + (dx - 7): the row taps live in
//     the transformed kernel (row -1 wraps onto the empty slot 11), every window is ONE row of a GEMM;
//   * channel pairs ride one complex transform, z_p = a_2p + i a_2p+1; per frequency pair (f, 1152 - f) the layer is a
//     real GEMM with K = N = 128 = [Z_p[f], Z_p[1152 - f]] x 32 pairs x (re, im), M = windows, in the split-fp16
//     arithmetic of the other convolution forms (h + l 2^-11, three f16 MFMAs per product block, f32 accumulate);
//   * 1152 = 24 x 48 = 24 x (2 x 24): a thread holds 24 complex points, two register-resident 24-point transforms
//     around ONE LDS transposition, the radix-2 step of the 48-point side is an exchange between lane pairs (DPP);
//     with the row pitch 96 = 2 x 48 a thread's points 48 n1 + n2 sit in image row n1 / 2 at column 48 (n1 & 1) + n2:
//     validity masks and addresses are compile-time per register;
//   * inverse transform -> BN + sigmoid (+ shortcut + BN) -> zero the gaps -> forward transform is one kernel per layer,
//     as in amt_fftconv.hip: between two such layers the activations exist in HBM only in the frequency domain, plus in
//     the spatial domain where a later shortcut reads them.
//
// Frequency tensors Xf / Yf [b][fp = 0 .. 576][128] f32, the 128 = pair group (4) x side (f | 1152 - f) x pair (8) x
// (re, im): 295 KB per window and tensor (spatial: 164 KB).  A row workgroup = (window, group of 8 pairs), 384 threads,
// two per CU (81 856 B of LDS each: 8 x 1152 complex + 4 dwords of pad per pair, and the first 952 twiddles; the few
// larger exponents use w^(j) = -w^(j - 576)).
#include "amt_fftconv_dev.h"
#include "amt_fftconv.h"
#include <algorithm>
#include <vector>

#define PK_NF 1152
#define PK_NP 577
#define PK_H 10
#define PK_W 64
#define PK_C 64
#define PK_KF 128                       // floats per (window, frequency pair)
#define PK_PS 2308                      // dwords per channel pair in the transposition buffer (1152 complex + 4 pad: at most
                                        // two-way bank conflicts on all four access shapes)
#define PK_TWN 952                      // twiddle entries in LDS
#define PK_THREADS 384                  // 8 pairs x 48

// a * conj(b)
__device__ __forceinline__ float2 pk_cmulc(float2 a, float2 b) {
    const amt_v2 t = amt_v2{a.y, a.y} * amt_v2{b.y, b.x};
    return f2(amt_v2{a.x, a.x} * amt_v2{b.x, -b.y} + t);
}
// the value of lane ^ 8 (row_ror:8 inside a row of 16 lanes)
__device__ __forceinline__ float pk_xor8(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true));
}
// e^{-2 pi i idx / 1152} from the LDS table of PK_TWN entries; BIG: idx may exceed the table
template <bool BIG>
__device__ __forceinline__ float2 pk_tw(const float2 *tw, int idx) {
    if (!BIG) return tw[idx];
    const bool big = idx >= PK_TWN;
    const float2 w = tw[big ? idx - PK_NF / 2 : idx];
    return big ? make_float2(-w.x, -w.y) : w;
}

struct PkRowArgs {
    const float *in_sp; size_t in_stride;        // [B][10][64][64] spatial input (first layer of a chain), or null
    const float *Yf;                             // [B][577][128] products of the GEMM, or null
    float *out_sp; size_t out_stride;            // spatial output, or null
    float *Xf;                                   // [B][577][128] transform of the output for the next layer, or null
    float *amaxf;                                // [B] max |Xf| per window (atomicMax, zeroed by the caller), with Xf
    float *amax_out;                             // [B] max |spatial output| per window, or null
    const float *s1, *t1, *s2, *t2;              // folded BN (+ conv bias) [64]; s2 / t2 null without a residual
    const float *sc; size_t sc_stride;           // shortcut tensor [B][10][64][64], or null
    const float2 *tw;                            // [PK_TWN] e^{-2 pi i m / 1152}
    int B;
};

// EPI (IN_FREQ only): 1 = no shortcut, no spatial output; 2 = shortcut tensor + spatial output; 4 = spatial output only.
template <bool IN_FREQ, int EPI>
__global__ __launch_bounds__(PK_THREADS, 4) void pk_row_kernel(PkRowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pk_smem[];
    float *buf = pk_smem;                                   // [8][PK_PS]
    float2 *tw = reinterpret_cast<float2 *>(pk_smem + 8 * PK_PS);
    const int tid = threadIdx.x;
    const int p8 = tid & 7, j48 = tid >> 3;
    // the four pair groups of a window run on one XCD (block ids 8 apart), so that its L2 assembles the 256-byte lines of
    // the spatial tensors they share
    const int blk = blockIdx.x;
    const int b = 8 * (blk >> 5) + (blk & 7), pg = (blk >> 3) & 3;
    if (b >= a.B) return;
    for (int i = tid; i < PK_TWN; i += PK_THREADS) tw[i] = a.tw[i];
    const int c0 = pg * 16 + p8 * 2;                        // first channel of the thread's pair
    const unsigned int kf_lane = (unsigned)((pg * 32 + p8 * 2) * 4);     // byte offset of the pair inside a 128-float block, side 0
    float2 x[24];
    if (IN_FREQ) {
        // ---- inverse, first half: thread (pair, k1, r) gathers W[k1 + 24 r + 48 m], transforms over m -> n', and the lane
        // pair (r = 0, 1) combines its two 24-point results into the 48 values D[n' + 24 s]
        int k1 = j48 >> 1, r = j48 & 1;
        const int f0 = k1 + 24 * r;
        const float *yb = a.Yf + (size_t)b * PK_NP * PK_KF;
        const unsigned int vlo = (unsigned)(f0 * PK_KF * 4) + kf_lane;                     // f = f0 + 48 m <= 576: side 0, pair fp = f
        const unsigned int vhi = (unsigned)((PK_NF - f0) * PK_KF * 4) + kf_lane + 64;      // else side 1, pair fp = 1152 - f
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const unsigned int step = (unsigned)(m * 48 * PK_KF * 4);
            unsigned int off = m < 12 ? vlo + step : vhi - step;
            if (m == 12) off = f0 == 0 ? vlo + step : vhi - step;                         // f = 576 + f0: self-paired for f0 = 0
            x[m] = fc_at<float2>(yb, off);
        }
        fc_fft24<true>(x);
        __syncthreads();                                    // the twiddle table is in place
        {
            const float sgn = r ? -1.0f : 1.0f;
            int idx = r ? 24 * k1 : 0;                      // k1 (n' + 24 r)
            const int wsel = r ? 24 : 0;
            float *dst = buf + p8 * PK_PS + 2 * (k1 * 48 + 24 * r);
#pragma unroll
            for (int n = 0; n < 24; ++n) {
                const float2 mine = pk_cmulc(x[n], tw[wsel * n]);                         // conj(W_48^n') on the odd half, 1 on the even
                const float2 other = make_float2(pk_xor8(mine.x), pk_xor8(mine.y));
                float2 d = make_float2(fmaf(sgn, mine.x, other.x), fmaf(sgn, mine.y, other.y));
                d = n >= 18 ? pk_cmulc(d, pk_tw<true>(tw, idx)) : pk_cmulc(d, pk_tw<false>(tw, idx));
                idx += k1;
                *reinterpret_cast<float2 *>(dst + 2 * n) = d;
            }
        }
        __syncthreads();
        // ---- second half: thread (pair, n2 = j48) transforms over k1 -> n1: a[48 n1 + n2]
        int n2 = j48;
        asm volatile("" : "+v"(n2));
#pragma unroll
        for (int k = 0; k < 24; ++k) x[k] = *reinterpret_cast<const float2 *>(buf + p8 * PK_PS + 2 * (k * 48 + n2));
        fc_fft24<true>(x);
        // ---- epilogue on the registers: register n1 holds image row n1 / 2, column 48 (n1 & 1) + n2 (valid: n1 < 20 and
        // column < 64); everything else of the sequence is the zero padding the next transform needs
        const float inv_n = 1.0f / (float)PK_NF;
        // sigmoid(x / N * s1 + t1) = 1 / (1 + 2^(x k1 + k0))
        const float k1a = -a.s1[c0] * (inv_n * FC_LOG2E), k1b = -a.s1[c0 + 1] * (inv_n * FC_LOG2E);
        const float k0a = -a.t1[c0] * FC_LOG2E, k0b = -a.t1[c0 + 1] * FC_LOG2E;
        const bool odd_ok = n2 < PK_W - 48;
        if constexpr (EPI == 1) {
#pragma unroll
            for (int n1 = 0; n1 < 24; ++n1) {
                if (n1 >= 2 * PK_H) { x[n1] = make_float2(0.f, 0.f); continue; }
                float2 v;
                v.x = fc_sigmoid_affine(x[n1].x, k1a, k0a);
                v.y = fc_sigmoid_affine(x[n1].y, k1b, k0b);
                if ((n1 & 1) && !odd_ok) v = make_float2(0.f, 0.f);
                x[n1] = v;
            }
        } else {
            constexpr bool RES = EPI == 2;
            const float s2a = RES ? a.s2[c0] : 1.f, s2b = RES ? a.s2[c0 + 1] : 1.f;
            const float t2a = RES ? a.t2[c0] : 0.f, t2b = RES ? a.t2[c0 + 1] : 0.f;
            const float *scb = RES ? a.sc + (size_t)b * a.sc_stride : nullptr;
            float *ob = a.out_sp + (size_t)b * a.out_stride;
            // byte offset of (column, channel pair) inside an image row: even registers column n2, odd ones 48 + n2
            // (clamped to the row: lanes beyond column 63 are masked, their address must still be inside the tensor)
            const unsigned int ve = (unsigned)((n2 * PK_C + c0) * 4);
            const unsigned int vo = (unsigned)(((48 + min(n2, PK_W - 49)) * PK_C + c0) * 4);
            constexpr int NB = 4;
            float2 scv[2][NB];
            auto request = [&](int batch, float2 (&dst)[NB]) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int n1 = batch * NB + i;
                    dst[i] = fc_at<float2>(scb + (n1 >> 1) * PK_W * PK_C, (n1 & 1) ? vo : ve);
                }
            };
            float vmax = 0.f;
            if constexpr (RES) request(0, scv[0]);
#pragma unroll
            for (int batch = 0; batch < 2 * PK_H / NB; ++batch) {
                if constexpr (RES) {
                    if (batch + 1 < 2 * PK_H / NB) request(batch + 1, scv[(batch + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int n1 = batch * NB + i;
                    float2 v;
                    v.x = fc_sigmoid_affine(x[n1].x, k1a, k0a);
                    v.y = fc_sigmoid_affine(x[n1].y, k1b, k0b);
                    if constexpr (RES) {
                        v.x = (v.x + scv[batch & 1][i].x) * s2a + t2a;
                        v.y = (v.y + scv[batch & 1][i].y) * s2b + t2b;
                    }
                    const bool ok = !(n1 & 1) || odd_ok;
                    if (!ok) v = make_float2(0.f, 0.f);
                    vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    if (ok) fc_at<float2>(ob + (n1 >> 1) * PK_W * PK_C, (n1 & 1) ? vo : ve) = v;
                    x[n1] = v;
                }
                if constexpr (RES) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n1 = 2 * PK_H; n1 < 24; ++n1) x[n1] = make_float2(0.f, 0.f);
            if (a.amax_out) {
                vmax = wave_max(vmax);
                if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(a.amax_out) + b, __float_as_int(vmax));
            }
            if (!a.Xf) return;
        }
        __syncthreads();                                    // every thread has read its column: `buf` is free again
    } else {
        const int n2 = j48;
        const float *ib = a.in_sp + (size_t)b * a.in_stride;
        const unsigned int ve = (unsigned)((n2 * PK_C + c0) * 4);
        const unsigned int vo = (unsigned)(((48 + min(n2, PK_W - 49)) * PK_C + c0) * 4);
#pragma unroll
        for (int n1 = 0; n1 < 2 * PK_H; ++n1) x[n1] = fc_at<float2>(ib + (n1 >> 1) * PK_W * PK_C, (n1 & 1) ? vo : ve);
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1)
            if (n1 >= 2 * PK_H || ((n1 & 1) && n2 >= PK_W - 48)) x[n1] = make_float2(0.f, 0.f);
        __syncthreads();                                    // the twiddle table is in place
    }
    // ---- forward: thread (pair, n2 = j48) transforms over n1 -> k1, twiddles W^{n2 k1}, transposition; thread (pair, k1, r)
    // transforms the half n2 = 2 m + r over m -> k2', the lane pair combines: X[k1 + 24 k2' + 576 s]
    {
        int n2 = j48, pq = p8;
        asm volatile("" : "+v"(n2), "+v"(pq));              // (opaque copies: see fc_row_kernel)
        fc_fft24<false>(x);
        int idx = 0;
        float *dst = buf + pq * PK_PS + 2 * n2;
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) {
            const float2 v = k1 >= 21 ? cmul(x[k1], pk_tw<true>(tw, idx)) : cmul(x[k1], pk_tw<false>(tw, idx));
            idx += n2;
            *reinterpret_cast<float2 *>(dst + 2 * 48 * k1) = v;
        }
        __syncthreads();
        int jj = j48;
        asm volatile("" : "+v"(jj), "+v"(pq));
        const int k1 = jj >> 1, r = jj & 1;
        const float *src = buf + pq * PK_PS + 2 * (k1 * 48 + r);
#pragma unroll
        for (int m = 0; m < 24; ++m) x[m] = *reinterpret_cast<const float2 *>(src + 4 * m);
        fc_fft24<false>(x);
        const float sgn = r ? -1.0f : 1.0f;
        const int wsel = r ? 24 : 0;
        float fmax_ = 0.f;
        float *xb = a.Xf + (size_t)b * PK_NP * PK_KF;
        // r = 0: f = k1 + 24 k2' < 576: pair f, side 0;  r = 1: f = 576 + k1 + 24 k2': pair 576 - k1 - 24 k2', side 1
        // (f = 576 itself: pair 576, side 0)
        const unsigned int kl = (unsigned)((pg * 32 + pq * 2) * 4);
        unsigned int off = r ? (unsigned)((PK_NF / 2 - k1) * PK_KF * 4) + kl + 64 : (unsigned)(k1 * PK_KF * 4) + kl;
        const unsigned int step = (unsigned)(24 * PK_KF * 4);
#pragma unroll
        for (int k2 = 0; k2 < 24; ++k2) {
            const float2 mine = cmul(x[k2], tw[wsel * k2]);
            const float2 other = make_float2(pk_xor8(mine.x), pk_xor8(mine.y));
            const float2 v = make_float2(fmaf(sgn, mine.x, other.x), fmaf(sgn, mine.y, other.y));
            if (k2 == 0) {
                // the self-paired bins f = 0 (r = 0, k1 = 0) and f = 576 (r = 1, k1 = 0) live on side 0; their side 1 is
                // not used by the GEMM (zero weights) but must hold finite numbers
                if (k1 == 0) {
                    const unsigned int o0 = (unsigned)((r ? PK_NF / 2 : 0) * PK_KF * 4) + kl;
                    fc_at<float2>(xb, o0) = v;
                    fc_at<float2>(xb, o0 + 64) = make_float2(0.f, 0.f);
                } else {
                    fc_at<float2>(xb, off) = v;
                }
            } else {
                fc_at<float2>(xb, off) = v;
            }
            off = r ? off - step : off + step;
            fmax_ = fmaxf(fmax_, fmaxf(fabsf(v.x), fabsf(v.y)));
        }
        fmax_ = wave_max(fmax_);
        if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(a.amaxf) + b, __float_as_int(fmax_));
    }
}

// ---------------------------------------------------------------------------------------------
// Per-frequency-pair GEMM: Yf[b][fp][n] = sum_k Xf[b][fp][k] G[fp][k][n], K = N = 128, M = windows.
// 256 threads: wave wn owns output columns 32 wn .. 32 wn + 31 and keeps the pair's weight fragments for them (4 k-steps
// x 2 N-tiles x h / l = 64 registers) across all the windows it is given; a chunk of 64 windows is split into f16 h / l
// planes in LDS (the next chunk's loads are in flight under the MFMAs).  The grid is persistent: 2 workgroups per CU,
// each walks a contiguous range of (pair, chunk) tasks, so a pair's weights are read once or twice in all.
// ---------------------------------------------------------------------------------------------
#define PK_CW 64                        // windows per chunk
#define PK_APITCH 144                   // halfs per staged row (128 + 16 pad = eighteen 16-byte slots: conflict-free fragment reads)
#define PK_OPITCH 36

struct PkGemmArgs {
    const float *Xf; float *Yf;
    const float *amaxf;                  // [B]
    const _Float16 *gw;                  // [577][4 k-steps][2 planes][128 n][32 k]
    const int *gsw;                      // [577] weights of pair fp were scaled by 2^gsw
    int B, ntasks, tasks_per_wg, nchunks;
};

__global__ __launch_bounds__(256, 2) void pk_gemm_kernel(PkGemmArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 ah[PK_CW * PK_APITCH];
    __shared__ __attribute__((aligned(16))) _Float16 al[PK_CW * PK_APITCH];
    __shared__ __attribute__((aligned(16))) float patch[4 * 16 * PK_OPITCH];
    __shared__ float sa_s[2 * PK_CW];
    const int tid = threadIdx.x, wn = tid >> 6, lane = tid & 63;
    const int t0 = blockIdx.x * a.tasks_per_wg, t1 = min(t0 + a.tasks_per_wg, a.ntasks);
    if (t0 >= t1) return;
    fc_h8 bh[4][2], bl[4][2];
    int sw = 0, fp_loaded = -1;
    constexpr int NLD = PK_CW * PK_KF / 4 / 256;            // float4 per thread and chunk: 8
    fc_f4 pre[NLD];
    auto fetch = [&](int task) {
        const int fp = task / a.nchunks, b0 = (task - fp * a.nchunks) * PK_CW;
        const int rows = min(PK_CW, a.B - b0);
        const unsigned char *base = reinterpret_cast<const unsigned char *>(a.Xf + ((size_t)b0 * PK_NP + fp) * PK_KF);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + 256 * u, row = min(i >> 5, rows - 1);
            pre[u] = *reinterpret_cast<const fc_f4 *>(base + (size_t)row * (PK_NP * PK_KF * 4) + (i & 31) * 16);
        }
    };
    fetch(t0);
    for (int task = t0; task < t1; ++task) {
        const int fp = task / a.nchunks, b0 = (task - fp * a.nchunks) * PK_CW;
        const int rows = min(PK_CW, a.B - b0);
        if (fp != fp_loaded) {
            const _Float16 *g = a.gw + (size_t)fp * (4 * 2 * PK_KF * 32);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = 16 * (2 * wn + nt) + (lane & 15);
                    bh[ks][nt] = *reinterpret_cast<const fc_h8 *>(g + ((size_t)(ks * 2 + 0) * PK_KF + n) * 32 + 8 * (lane >> 4));
                    bl[ks][nt] = *reinterpret_cast<const fc_h8 *>(g + ((size_t)(ks * 2 + 1) * PK_KF + n) * 32 + 8 * (lane >> 4));
                }
            sw = a.gsw[fp];
            fp_loaded = fp;
        }
        __syncthreads();                                    // the previous chunk's fragments have been read
        if (tid < PK_CW) {
            const int s_ = tid < rows ? fc_scale_exp(a.amaxf[b0 + tid]) : 0;
            sa_s[tid] = ldexpf(1.0f, s_);
            sa_s[PK_CW + tid] = ldexpf(1.0f, -(s_ + sw));
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + 256 * u, row = i >> 5, c4 = i & 31;
            const fc_f4 v = pre[u] * sa_s[row];
            const fc_h4 hh = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const fc_h4 ll = {(_Float16)((v.x - (float)hh.x) * FC_LSCALE), (_Float16)((v.y - (float)hh.y) * FC_LSCALE),
                              (_Float16)((v.z - (float)hh.z) * FC_LSCALE), (_Float16)((v.w - (float)hh.w) * FC_LSCALE)};
            *reinterpret_cast<fc_h4 *>(ah + row * PK_APITCH + 4 * c4) = hh;
            *reinterpret_cast<fc_h4 *>(al + row * PK_APITCH + 4 * c4) = ll;
        }
        __syncthreads();
        if (task + 1 < t1) fetch(task + 1);                 // travels while this chunk is multiplied
        float *pt = patch + wn * 16 * PK_OPITCH;
        unsigned char *ybase = reinterpret_cast<unsigned char *>(a.Yf + ((size_t)b0 * PK_NP + fp) * PK_KF);
        const int n_mt = (rows + 15) >> 4;
        for (int mt = 0; mt < n_mt; ++mt) {
            const _Float16 *pa = ah + (16 * mt + (lane & 15)) * PK_APITCH + 8 * (lane >> 4);
            const _Float16 *pl = al + (16 * mt + (lane & 15)) * PK_APITCH + 8 * (lane >> 4);
            fc_f4 hi[2], lo[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { hi[nt] = fc_f4{0.f, 0.f, 0.f, 0.f}; lo[nt] = fc_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const fc_h8 fa = *reinterpret_cast<const fc_h8 *>(pa + 32 * ks);
                const fc_h8 fl = *reinterpret_cast<const fc_h8 *>(pl + 32 * ks);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, bh[ks][nt], hi[nt], 0, 0, 0);
                    lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, bl[ks][nt], lo[nt], 0, 0, 0);
                    lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, bh[ks][nt], lo[nt], 0, 0, 0);
                }
            }
            // D (column n = lane & 15, rows 4 (lane >> 4) + e) through the wave's LDS patch: every lane stores 16 bytes, a
            // window's 32 columns leave as one 128-byte line
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pt[(4 * (lane >> 4) + e) * PK_OPITCH + 16 * nt + (lane & 15)] = hi[nt][e] + lo[nt][e] * (1.0f / FC_LSCALE);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rr = 8 * half + (lane >> 3), ro = 16 * mt + rr;
                fc_f4 v = *reinterpret_cast<const fc_f4 *>(pt + rr * PK_OPITCH + 4 * (lane & 7));
                if (ro < rows) {
                    v *= sa_s[PK_CW + ro];
                    *reinterpret_cast<fc_f4 *>(ybase + (size_t)ro * (PK_NP * PK_KF * 4) + (32 * wn + 4 * (lane & 7)) * 4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Transformed kernel matrices on the device, in float64 (one workgroup per frequency pair):
//   Kf[fo][ci][co] = sum_{dy, dx} K[dy][dx][ci][co] e^{+2 pi i fo ((dy - 1) 96 + (dx - 7)) / 1152},
// folded with the (de)interleaving of the packed channel pairs into the real 128 x 128 matrix G[fp] of
// scripts/fftconv_packed_model.py (pair_matrices), scaled per pair to max |G| 2^sw in [8, 16) and split into f16 h / l.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pk_weights_kernel(const float *__restrict__ kern /* [4][16][64][64] */, _Float16 *gw, int *gsw) {
    extern __shared__ __attribute__((aligned(16))) double pkw_smem[];
    double *G = pkw_smem;                                   // [128][128]
    double *ph = G + 128 * 128;                             // [2][64][2] (cos, sin)
    double *red = ph + 2 * 64 * 2;                          // [4]
    const int fp = blockIdx.x, tid = threadIdx.x;
    const bool self = fp == 0 || fp == PK_NF / 2;
    if (tid < 128) {
        const int t = tid >> 6, tap = tid & 63, dy = tap >> 4, dx = tap & 15;
        const int fo = t == 0 ? fp : (PK_NF - fp) % PK_NF;
        const int off = (dy - 1) * 96 + (dx - 7);
        const int m = (int)((((long)fo * off) % PK_NF + PK_NF) % PK_NF);
        double s, c;
        sincos(6.283185307179586476925286766559 * (double)m / (double)PK_NF, &s, &c);
        ph[(t * 64 + tap) * 2] = c; ph[(t * 64 + tap) * 2 + 1] = s;
    }
    for (int i = tid; i < 128 * 128; i += 256) G[i] = 0.0;
    __syncthreads();
    // model index (side, pair p, c) -> tensor index pg * 32 + side * 16 + p8 * 2 + c
    auto kidx = [](int side, int p, int c) { return (p >> 3) * 32 + side * 16 + (p & 7) * 2 + c; };
    for (int pq = tid; pq < 32 * 32; pq += 256) {
        const int p = pq >> 5, q = pq & 31;
        for (int t = 0; t < 2; ++t) {
            // e = Kf[2p][.], o = Kf[2p + 1][.] at output channels 2q, 2q + 1
            double er0 = 0, ei0 = 0, er1 = 0, ei1 = 0, or0 = 0, oi0 = 0, or1 = 0, oi1 = 0;
            for (int tap = 0; tap < 64; ++tap) {
                const double c = ph[(t * 64 + tap) * 2], s = ph[(t * 64 + tap) * 2 + 1];
                const float *kp = kern + (size_t)tap * PK_C * PK_C;
                const double k00 = kp[(2 * p) * PK_C + 2 * q], k01 = kp[(2 * p) * PK_C + 2 * q + 1];
                const double k10 = kp[(2 * p + 1) * PK_C + 2 * q], k11 = kp[(2 * p + 1) * PK_C + 2 * q + 1];
                er0 += k00 * c; ei0 += k00 * s; er1 += k01 * c; ei1 += k01 * s;
                or0 += k10 * c; oi0 += k10 * s; or1 += k11 * c; oi1 += k11 * s;
            }
            // cz[c] = (e - i o) / 2, cc[c] = (e + i o) / 2;  wz = cz[2q] + i cz[2q+1], wc = cc[2q] + i cc[2q+1]
            const double cz0r = 0.5 * (er0 + oi0), cz0i = 0.5 * (ei0 - or0), cz1r = 0.5 * (er1 + oi1), cz1i = 0.5 * (ei1 - or1);
            const double cc0r = 0.5 * (er0 - oi0), cc0i = 0.5 * (ei0 + or0), cc1r = 0.5 * (er1 - oi1), cc1i = 0.5 * (ei1 + or1);
            const double wzr = cz0r - cz1i, wzi = cz0i + cz1r, wcr = cc0r - cc1i, wci = cc0i + cc1r;
            const int s_z = t == 0 ? 0 : 1, s_c = 1 - s_z;
            const int kz0 = kidx(s_z, p, 0), kz1 = kidx(s_z, p, 1), kc0 = kidx(s_c, p, 0), kc1 = kidx(s_c, p, 1);
            const int n0 = kidx(t, q, 0), n1 = kidx(t, q, 1);
            G[kz0 * 128 + n0] = wzr; G[kz1 * 128 + n0] = -wzi; G[kz0 * 128 + n1] = wzi; G[kz1 * 128 + n1] = wzr;
            G[kc0 * 128 + n0] = wcr; G[kc1 * 128 + n0] = wci; G[kc0 * 128 + n1] = wci; G[kc1 * 128 + n1] = -wcr;
        }
    }
    __syncthreads();
    if (self) {
        // f = -f: both sides of the input are the same bin; keep side 0 (the row kernel writes zeros to side 1)
        for (int i = tid; i < 64 * 128; i += 256) {
            const int k = i >> 7, n = i & 127;              // k runs over (pg, p8, c) of side 0
            const int k0 = (k >> 4) * 32 + (k & 15), ks1 = k0 + 16;
            const bool n_side1 = (n >> 4) & 1;
            const double v = n_side1 ? 0.0 : G[k0 * 128 + n] + G[ks1 * 128 + n];
            G[k0 * 128 + n] = v;
            G[ks1 * 128 + n] = 0.0;
        }
        __syncthreads();
    }
    double m = 0.0;
    for (int i = tid; i < 128 * 128; i += 256) m = fmax(m, fabs(G[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    int sw = 0;
    if (m > 0) { int e; (void)frexp(m, &e); sw = 4 - e; }   // max |G| 2^sw in [8, 16)
    if (tid == 0) gsw[fp] = sw;
    const double scl = ldexp(1.0, sw);
    _Float16 *g = gw + (size_t)fp * (4 * 2 * PK_KF * 32);
    for (int i = tid; i < 128 * 128; i += 256) {
        const int k = i >> 7, n = i & 127;
        const float v = (float)(G[i] * scl);
        _Float16 hh = (_Float16)v;
        if (!(fabsf(v) >= 6.103515625e-05f)) hh = (_Float16)0.0f;
        const float rr = (v - (float)hh) * FC_LSCALE;
        _Float16 ll = (_Float16)rr;
        if (!(fabsf(rr) >= 6.103515625e-05f)) ll = (_Float16)0.0f;
        const int ks = k >> 5, kk = k & 31;
        g[((size_t)(ks * 2 + 0) * PK_KF + n) * 32 + kk] = hh;
        g[((size_t)(ks * 2 + 1) * PK_KF + n) * 32 + kk] = ll;
    }
}

// =====================================================================================
// Host side
// =====================================================================================
struct amt_fftpk_layer {
    _Float16 *gw = nullptr;
    int *gsw = nullptr;
    float2 *tw = nullptr;
    float *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;      // optional device copies (stand-alone entry)
};

void amt_fftpk_layer_destroy_internal(amt_fftpk_layer *L) {
    if (!L) return;
    for (void *p : {(void *)L->gw, (void *)L->gsw, (void *)L->tw, (void *)L->s1, (void *)L->t1, (void *)L->s2, (void *)L->t2})
        if (p) (void)hipFree(p);
    delete L;
}

int amt_fftpk_layer_create_internal(amt_fftpk_layer **out, const float *kernel /* [4][16][64][64] host */) {
    if (!out || !kernel) return AMT_E_INVALID;
    amt_fftpk_layer *L = new amt_fftpk_layer();
    const size_t kbytes = (size_t)4 * 16 * PK_C * PK_C * sizeof(float);
    const size_t gw_halfs = (size_t)PK_NP * 4 * 2 * PK_KF * 32;
    std::vector<float2> tw(PK_TWN);
    for (int m = 0; m < PK_TWN; ++m) {
        const double ang = -6.283185307179586476925286766559 * m / PK_NF;
        tw[m] = make_float2((float)cos(ang), (float)sin(ang));
    }
    float *kdev = nullptr;
    hipError_t e = hipMalloc(&kdev, kbytes);
    if (e == hipSuccess) e = hipMemcpy(kdev, kernel, kbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&L->gw, gw_halfs * sizeof(_Float16));
    if (e == hipSuccess) e = hipMalloc(&L->gsw, PK_NP * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&L->tw, tw.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(L->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t lds = (size_t)(128 * 128 + 2 * 64 * 2 + 4) * sizeof(double);
        e = hipFuncSetAttribute((const void *)pk_weights_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            pk_weights_kernel<<<PK_NP, 256, lds, 0>>>(kdev, L->gw, L->gsw);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(0);
    }
    if (kdev) (void)hipFree(kdev);
    if (e != hipSuccess) {
        snprintf(amt_hip_err_buf, sizeof(amt_hip_err_buf), "fftpk weights: %s", hipGetErrorString(e));
        amt_fftpk_layer_destroy_internal(L);
        return AMT_E_HIP;
    }
    *out = L;
    return AMT_OK;
}

size_t amt_fftpk_freq_floats(int B) { return (size_t)PK_NP * PK_KF * B; }

static const size_t PK_ROW_LDS = (size_t)(8 * PK_PS) * 4 + PK_TWN * sizeof(float2);
template <bool IN_FREQ, int EPI>
static int pk_row_launch_t(const PkRowArgs &a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)pk_row_kernel<IN_FREQ, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PK_ROW_LDS));
        attr = true;
    }
    pk_row_kernel<IN_FREQ, EPI><<<32 * ((a.B + 7) / 8), PK_THREADS, PK_ROW_LDS, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_fftpk_forward_fft(const amt_fftpk_layer *L, const float *in_sp, size_t in_stride, int B, float *Xf, float *amaxf, hipStream_t st) {
    if (!L || !in_sp || !Xf || !amaxf || B <= 0) return AMT_E_INVALID;
    AMT_HIP_CHECK(hipMemsetAsync(amaxf, 0, (size_t)B * sizeof(float), st));
    PkRowArgs a{};
    a.in_sp = in_sp; a.in_stride = in_stride; a.Xf = Xf; a.amaxf = amaxf; a.tw = L->tw; a.B = B;
    return pk_row_launch_t<false, 1>(a, st);
}

int amt_fftpk_gemm(const amt_fftpk_layer *L, const float *Xf, const float *amaxf, int B, float *Yf, hipStream_t st) {
    if (!L || !Xf || !Yf || !amaxf || B <= 0) return AMT_E_INVALID;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    PkGemmArgs a{Xf, Yf, amaxf, L->gw, L->gsw, B, 0, 0, 0};
    a.nchunks = (B + PK_CW - 1) / PK_CW;
    a.ntasks = PK_NP * a.nchunks;
    const int wgs = std::min(a.ntasks, 2 * cus);
    a.tasks_per_wg = (a.ntasks + wgs - 1) / wgs;
    pk_gemm_kernel<<<(a.ntasks + a.tasks_per_wg - 1) / a.tasks_per_wg, 256, 0, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_fftpk_inverse_epilogue(const amt_fftpk_layer *L, const float *Yf, const FcEpilogue &ep, int B, float *out_sp, size_t out_stride,
                               float *Xf_next, float *amaxf_next, float *amax_out, hipStream_t st) {
    if (!L || !Yf || !ep.s1 || !ep.t1 || (!out_sp && !Xf_next) || B <= 0) return AMT_E_INVALID;
    if (Xf_next && !amaxf_next) return AMT_E_INVALID;
    if (ep.sc1) return AMT_E_UNSUPPORTED;
    if (Xf_next) AMT_HIP_CHECK(hipMemsetAsync(amaxf_next, 0, (size_t)B * sizeof(float), st));
    PkRowArgs a{};
    a.Yf = Yf; a.out_sp = out_sp; a.out_stride = out_stride; a.Xf = Xf_next; a.amaxf = amaxf_next; a.amax_out = amax_out;
    a.s1 = ep.s1; a.t1 = ep.t1; a.s2 = ep.s2; a.t2 = ep.t2; a.sc = ep.sc; a.sc_stride = ep.sc_stride;
    a.tw = L->tw; a.B = B;
    if (!out_sp) {
        if (a.sc || a.amax_out) return AMT_E_UNSUPPORTED;
        return pk_row_launch_t<true, 1>(a, st);
    }
    if (a.sc) {
        if (!a.s2 || !a.t2) return AMT_E_INVALID;
        return pk_row_launch_t<true, 2>(a, st);
    }
    return pk_row_launch_t<true, 4>(a, st);
}

// ---- stand-alone C ABI entry (tests / microbenchmarks): one layer, spatial in, spatial out --------------------------
extern "C" {

int amt_fftpk_create(amt_fftpk_layer **layer, const float *kernel_host, const float *s1, const float *t1, const float *s2,
                     const float *t2) {
    if (!layer || !kernel_host || !s1 || !t1) return AMT_E_INVALID;
    int rc = amt_fftpk_layer_create_internal(layer, kernel_host);
    if (rc != AMT_OK) return rc;
    amt_fftpk_layer *L = *layer;
    auto up = [&](const float *h, float **d) -> hipError_t {
        if (!h) return hipSuccess;
        hipError_t e = hipMalloc(d, PK_C * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(*d, h, PK_C * sizeof(float), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(s1, &L->s1);
    if (e == hipSuccess) e = up(t1, &L->t1);
    if (e == hipSuccess) e = up(s2, &L->s2);
    if (e == hipSuccess) e = up(t2, &L->t2);
    if (e != hipSuccess) { amt_fftpk_layer_destroy_internal(L); *layer = nullptr; return AMT_E_HIP; }
    return AMT_OK;
}

int amt_fftpk_destroy(amt_fftpk_layer *layer) { amt_fftpk_layer_destroy_internal(layer); return AMT_OK; }

size_t amt_fftpk_workspace_bytes(int B) { return (2 * amt_fftpk_freq_floats(B) + 2 * (size_t)B) * sizeof(float); }

int amt_fftpk_run(const amt_fftpk_layer *L, const float *in, const float *shortcut, int B, float *out, void *workspace,
                  size_t workspace_bytes, int chain, int repeat_gemm, void *stream) {
    if (!L || !in || !out || !workspace || B <= 0) return AMT_E_INVALID;
    if (workspace_bytes < amt_fftpk_workspace_bytes(B)) return AMT_E_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    float *Xf = (float *)workspace, *Yf = Xf + amt_fftpk_freq_floats(B), *amaxf = Yf + amt_fftpk_freq_floats(B);
    const size_t stride = (size_t)PK_H * PK_W * PK_C;
    int rc = amt_fftpk_forward_fft(L, in, stride, B, Xf, amaxf, st);
    FcEpilogue ep{};
    ep.s1 = L->s1; ep.t1 = L->t1;
    // chain > 0: that many extra applications of the same layer with the activations handed over in the frequency domain
    // (no shortcut, no spatial output: the register epilogue and the re-zeroing of the gaps), then the last one as below
    for (int c = 0; c < chain && rc == AMT_OK; ++c) {
        rc = amt_fftpk_gemm(L, Xf, amaxf, B, Yf, st);
        if (rc == AMT_OK) rc = amt_fftpk_inverse_epilogue(L, Yf, ep, B, nullptr, 0, Xf, amaxf, nullptr, st);
    }
    for (int i = 0; i < (repeat_gemm > 0 ? repeat_gemm : 1) && rc == AMT_OK; ++i) rc = amt_fftpk_gemm(L, Xf, amaxf, B, Yf, st);
    if (rc != AMT_OK) return rc;
    if (shortcut && L->s2) { ep.s2 = L->s2; ep.t2 = L->t2; ep.sc = shortcut; ep.sc_stride = stride; }
    return amt_fftpk_inverse_epilogue(L, Yf, ep, B, out, stride, nullptr, nullptr, nullptr, st);
}

}  // extern "C"
