// FFT-domain form of the timing head's 4 x 16 convolutions on its large images (conv mode 3, experimental).
//
// Replaces, for the layers it is built for, Conv2D(32 -> 32, (4, 16), 'same') + BatchNormalization + sigmoid
// (+ Add + BatchNormalization) of /root/reference/RDCNN.py:186-193 on the 20 x 516 (N = 2048) / 20 x 258 (N = 4096)
// images of timing_classifier.py:13-36 -- the eleven layers per timing net that are 63 % of a C3 step in the direct
// split-fp16 form (3.3 ms per 1024 windows and layer, at the chip's practical matrix-pipe ceiling: DESIGN 7).
// The kernels are 16 taps long in the time direction; transformed along it, a layer needs 7 x fewer multiplies:
//
//   * rows of W <= 561 positions are transformed by 576-point COMPLEX FFTs of channel PAIRS z_p = a_2p + i a_2p+1
//     (no real-FFT separation pass; 576 = 24 x 24: two register-resident 24-point transforms per thread around ONE
//     LDS transposition);
//   * per frequency pair (f, 576 - f) the layer is one real GEMM, K = 4 row taps x [Z_p[f], Z_p[576 - f]] = 256,
//     N = [W_q[f], W_q[576 - f]] = 64, on the f16 matrix pipe in the same split-fp16 arithmetic as the direct form
//     (h + l 2^-11, three MFMAs per product block, hi / lo accumulators); the (de)interleaving of the packed channels
//     lives in the transformed kernel matrices (host, float64; scripts/fftconv_model.py is the numpy model);
//   * the inverse transform of W_q = Y_2q + i Y_2q+1 returns two real output channels per sequence, already in the
//     register layout the next layer's forward transform starts from, so BN + sigmoid (+ shortcut + BN) and the next
//     forward transform run in the same kernel: between two such layers the activations never exist in HBM in the
//     spatial domain unless a later shortcut reads them.
//
// Frequency tensors: Xf / Yf [b][fp = 0 .. 288][h][64] f32, the 64 = side (f | 576 - f) x pair (16) x (re, im): a GEMM
// workgroup reads H x 256 contiguous bytes per (window, fp), a row transform 256-byte pieces H x 256 bytes apart inside
// its window's 1.5 MB (the first layout, [fp][b][h][64], put every piece of a row 5 MB from the next: 1.8 TB/s).
#include "amt_fftconv_dev.h"
#include "amt_fftconv.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#define FC_NF 576
#define FC_NP 289
#define FC_PS 1158                      // dwords per channel pair in the LDS transposition buffer (576 complex + 6 pad: the 64-bit
                                        // reads, 16 pairs x four columns 48 dwords apart per wave, then touch every bank pair once; 1160
                                        // made them four-way conflicts; the writes are two-way now, inside their issue time)
#define FC_THREADS 384                  // 16 channel pairs x 24

struct FcRowArgs {
    const float *in_sp; size_t in_stride;        // [B][H][W][32] spatial input (first layer of a chain), or null
    const float *Yf;                             // [B][289][H][64] products of the GEMM (all other layers), or null
    float *out_sp; size_t out_stride;            // spatial output [B][H][W][32], or null (no later reader)
    float *Xf;                                   // [B][289][H][64] transform of the output for the next layer, or null
    float *amaxf;                                // [B] max |Xf| per window (atomicMax, zeroed by the caller), with Xf
    float *amax_out;                             // [B] max |spatial output| per window, or null
    const float *s1, *t1, *s2, *t2;              // folded BN (+ conv bias); s2 / t2 null without a residual
    const float *sc; size_t sc_stride;           // identity shortcut tensor [B][H][W][32], or null
    const float *sc1; size_t sc1_stride;         // rank-1 shortcut: the one-channel network input [B][H][W] ...
    const float *sc1_w, *sc1_s, *sc1_t;          // ... its 1x1 kernel and folded BN [32]
    const float2 *tw;                            // [576] e^{-2 pi i m / 576}
    int B, H, W;
    int sf, sh;                                  // float strides of the frequency tensors: pair fp, image row h (window: 289 H 64)
};

// One 384-thread workgroup per image row (b, h); TWO such workgroups share a CU (<= 128 registers: 4 + 4 + 2 + 2 waves fit
// four per SIMD; 2 x 78.7 KB of LDS) and run their phases independently of each other.  IN_FREQ: inverse transform of Yf +
// epilogue; else the spatial input is loaded.  Then (a.Xf) the forward transform.
// The epilogue runs on the registers the inverse transform leaves (24 positions x one channel pair per thread: BN + sigmoid
// are per channel, the thread's pair is fixed) and the forward transform starts from them.  EPI = 1: no shortcut, no spatial
// output (every second layer of a chain).  EPI = 2: shortcut (identity tensor or the rank-1 projection of the one-channel
// input), spatial output and its per-window maximum as the arguments ask: the 24 shortcut values of a thread are requested
// in one burst after the transform (a wave's instruction covers four positions x 128 bytes = 512 contiguous bytes) and the
// outputs leave from the registers the same way.  (Round 3 sent that form through an LDS round trip -- position-major
// copy, compact loop, re-gather, three barriers: 16.5 us of a row's 37.)
template <bool IN_FREQ, int EPI>
__global__ __launch_bounds__(FC_THREADS, 4) void fc_row_kernel(FcRowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fc_smem[];
    const int tid = threadIdx.x;
    float *buf = fc_smem;                                   // [16][FC_PS]
    float2 *tw = reinterpret_cast<float2 *>(fc_smem + 16 * FC_PS);
    const int c16 = (tid >> 2) & 15;                        // channel pair
    const int j24 = (tid & 3) + 4 * (tid >> 6);             // 0 .. 23
    const int row = blockIdx.x;
    const int b = row / a.H, h = row - b * a.H;
    for (int i = tid; i < FC_NF; i += FC_THREADS) tw[i] = a.tw[i];
    float2 x[24];
    if (IN_FREQ) {
        // ---- inverse, first half: thread (q = c16, k1 = j24) gathers W_q[k1 + 24 k2], transforms over k2 -> n2
        const int k1 = j24;
        const float *yb = a.Yf + (size_t)b * FC_NP * a.H * 64 + (size_t)h * a.sh;
        // f = k1 + 24 k2 <= 288 for k2 <= 11 (pair f, side 0) and > 288 for k2 >= 13 (pair 576 - f, side 1); k2 = 12: f = 288
        // for k1 = 0 (side 0), else side 1.  One per-lane offset each way, the k2 steps are uniform: they go into the scalar
        // base (per element the lane computed fp x stride as a 64-bit multiply-add, two selects and a compare)
        const unsigned int vlo = (unsigned)((k1 * a.sf + 2 * c16) * 4);
        const unsigned int vhi = (unsigned)(((FC_NF - k1) * a.sf + 32 + 2 * c16) * 4);
        const int kstep = 24 * a.sf;                        // floats
#pragma unroll
        for (int k2 = 0; k2 < 24; ++k2) {
            if (k2 < 12) x[k2] = fc_at<float2>(yb + k2 * kstep, vlo);
            else if (k2 > 12) x[k2] = fc_at<float2>(yb - k2 * kstep, vhi);      // (the lane offset stays non-negative: vhi >= 553 strides)
            else x[k2] = fc_at<float2>(yb, k1 == 0 ? vlo + (unsigned)(12 * kstep * 4) : vhi - (unsigned)(12 * kstep * 4));   // (both offsets >= 0)
        }
        fc_fft24<true>(x);
        __syncthreads();                                    // the twiddle table is in place
        // the twiddles in batches of eight reads in flight (read one by one, each product waited out its own LDS round trip)
#pragma unroll
        for (int g0 = 0; g0 < 24; g0 += 8) {
            float2 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = tw[(g0 + i) * k1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 v = fc_cmulc(x[g0 + i], w[i]);
                *reinterpret_cast<float2 *>(buf + c16 * FC_PS + ((g0 + i) * 24 + k1) * 2) = v;
            }
        }
        __syncthreads();
        // ---- second half: thread (q, n2 = j24) transforms over k1 -> n1: y[24 n1 + n2]
        const int n2 = j24;
#pragma unroll
        for (int k = 0; k < 24; ++k) x[k] = *reinterpret_cast<const float2 *>(buf + c16 * FC_PS + (n2 * 24 + k) * 2);
        fc_fft24<true>(x);
        // ---- epilogue on the registers
        const float inv_n = 1.0f / (float)FC_NF;
        const int c0 = 2 * c16;
        // sigmoid(x / N * s1 + t1) = 1 / (1 + 2^(x k1 + k0))
        const float k1a = -a.s1[c0] * (inv_n * FC_LOG2E), k1b = -a.s1[c0 + 1] * (inv_n * FC_LOG2E);
        const float k0a = -a.t1[c0] * FC_LOG2E, k0b = -a.t1[c0 + 1] * FC_LOG2E;
        if constexpr (EPI == 1) {
#pragma unroll
            for (int n1 = 0; n1 < 24; ++n1) {
                float2 v;
                v.x = fc_sigmoid_affine(x[n1].x, k1a, k0a);
                v.y = fc_sigmoid_affine(x[n1].y, k1b, k0b);
                if (24 * n1 + n2 >= a.W) v = make_float2(0.f, 0.f);
                x[n1] = v;
            }
        } else {
            // EPI = 2: identity shortcut tensor; 3: rank-1 shortcut (the projected one-channel input); 4: none.  All write
            // the spatial output.
            constexpr bool RES = EPI == 2 || EPI == 3;
            const float s2a = RES ? a.s2[c0] : 1.f, s2b = RES ? a.s2[c0 + 1] : 1.f;
            const float t2a = RES ? a.t2[c0] : 0.f, t2b = RES ? a.t2[c0 + 1] : 0.f;
            const float *scb = EPI == 2 ? a.sc + (size_t)b * a.sc_stride + (size_t)h * a.W * 32 : nullptr;
            const float *sc1b = EPI == 3 ? a.sc1 + (size_t)b * a.sc1_stride + (size_t)h * a.W : nullptr;
            float *ob = a.out_sp + (size_t)b * a.out_stride + (size_t)h * a.W * 32;
            // (opaque copy: the access offsets below depend only on thread constants, and hipcc otherwise forms all of
            // them at the top of the kernel and carries them -- through scratch memory -- across the transforms)
            int n2e = n2;
            asm volatile("" : "+v"(n2e));
            const unsigned int voff = (unsigned)((n2e * 32 + c0) * 4), wlim = (unsigned)((a.W - 1) * 128 + c0 * 4);
            float pw0 = 0.f, pw1 = 0.f, ps0 = 0.f, ps1 = 0.f, pt0 = 0.f, pt1 = 0.f;
            if constexpr (EPI == 3) {
                pw0 = a.sc1_w[c0]; pw1 = a.sc1_w[c0 + 1]; ps0 = a.sc1_s[c0]; ps1 = a.sc1_s[c0 + 1];
                pt0 = a.sc1_t[c0]; pt1 = a.sc1_t[c0 + 1];
            }
            // the thread's 24 shortcut values arrive in batches of NB, one batch ahead of the one being used (clamped
            // addresses, no branch): all 24 at once are 48 registers beside the 48 of the row and spill
            constexpr int NB = 4;
            float2 scv[2][NB];
            auto request = [&](int batch, float2 (&dst)[NB]) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const unsigned int o = min(voff + (unsigned)(24 * (batch * NB + i) * 128), wlim);
                    if constexpr (EPI == 2) dst[i] = fc_at<float2>(scb, o);
                    if constexpr (EPI == 3) dst[i].x = fc_at<float>(sc1b, (o >> 7) * 4);
                }
            };
            float vmax = 0.f;
            if constexpr (RES) request(0, scv[0]);
#pragma unroll
            for (int batch = 0; batch < 24 / NB; ++batch) {
                if constexpr (RES) {
                    if (batch + 1 < 24 / NB) request(batch + 1, scv[(batch + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int n1 = batch * NB + i, w = 24 * n1 + n2e;
                    float2 v;
                    v.x = fc_sigmoid_affine(x[n1].x, k1a, k0a);
                    v.y = fc_sigmoid_affine(x[n1].y, k1b, k0b);
                    if constexpr (EPI == 2) {
                        v.x = (v.x + scv[batch & 1][i].x) * s2a + t2a;
                        v.y = (v.y + scv[batch & 1][i].y) * s2b + t2b;
                    }
                    if constexpr (EPI == 3) {
                        const float xi = scv[batch & 1][i].x;
                        v.x = (v.x + (fmaf(xi, pw0, 0.f) * ps0 + pt0)) * s2a + t2a;
                        v.y = (v.y + (fmaf(xi, pw1, 0.f) * ps1 + pt1)) * s2b + t2b;
                    }
                    if (w >= a.W) v = make_float2(0.f, 0.f);
                    vmax = fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y)));
                    if (w < a.W) fc_at<float2>(ob, voff + (unsigned)(24 * n1 * 128)) = v;
                    x[n1] = v;
                }
                if constexpr (RES) __builtin_amdgcn_sched_barrier(0);
            }
            if (a.amax_out) {
                vmax = wave_max(vmax);
                if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(a.amax_out) + b, __float_as_int(vmax));
            }
            if (!a.Xf) return;
        }
        __syncthreads();                                    // every thread has read its column: `buf` is free for the forward transposition
    } else {
        // branch-free: the load goes to a clamped address and a select zeroes the padding (a conditional load is a branch,
        // and 24 branches are 24 serialised round trips to memory)
        const int n2 = j24;
        const float *rowp = a.in_sp + (size_t)b * a.in_stride + (size_t)h * a.W * 32;
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1) {
            const int w = 24 * n1 + n2;
            x[n1] = fc_at<float2>(rowp, (unsigned)((min(w, a.W - 1) * 32 + 2 * c16) * 4));
        }
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1)
            if (24 * n1 + n2 >= a.W) x[n1] = make_float2(0.f, 0.f);
        __syncthreads();                                    // the twiddle table is in place
    }
    // ---- forward: thread (p = c16, n2 = j24) transforms over n1 -> k1, twiddles, transposition, (p, k1 = j24) over n2 -> k2
    {
        // (opaque copies here too: the twiddle and LDS addresses of this phase are the inverse transposition's, and hipcc
        // otherwise keeps all of them alive from there -- through scratch memory -- across the epilogue)
        int n2 = j24, cq = c16;
        asm volatile("" : "+v"(n2), "+v"(cq));
        fc_fft24<false>(x);
#pragma unroll
        for (int g0 = 0; g0 < 24; g0 += 8) {
            float2 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = tw[n2 * (g0 + i)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 v = cmul(x[g0 + i], w[i]);
                *reinterpret_cast<float2 *>(buf + cq * FC_PS + ((g0 + i) * 24 + n2) * 2) = v;
            }
        }
        __syncthreads();
        int k1 = j24, cp = c16;
        // (opaque copies: the 24 store offsets below depend only on thread constants, and hipcc otherwise computes them at
        // the top of the kernel and carries them -- through scratch memory -- across all four transforms)
        asm volatile("" : "+v"(k1), "+v"(cp));
#pragma unroll
        for (int n = 0; n < 24; ++n) x[n] = *reinterpret_cast<const float2 *>(buf + cp * FC_PS + (k1 * 24 + n) * 2);
        fc_fft24<false>(x);
        float fmax_ = 0.f;
        float *xb = a.Xf + (size_t)b * FC_NP * a.H * 64 + (size_t)h * a.sh;
        // (addresses as in the inverse's loads: side 0 for k2 <= 11, side 1 for k2 >= 13, k2 = 12 by k1; the self-paired
        // bins f = 0 and f = 288 -- k1 = 0 only -- fill both sides)
        const unsigned int vlo = (unsigned)((k1 * a.sf + 2 * cp) * 4);
        const unsigned int vhi = (unsigned)(((FC_NF - k1) * a.sf + 32 + 2 * cp) * 4);
        const int kstep = 24 * a.sf;
#pragma unroll
        for (int k2 = 0; k2 < 24; ++k2) {
            if (k2 < 12) fc_at<float2>(xb + k2 * kstep, vlo) = x[k2];
            else if (k2 > 12) fc_at<float2>(xb - k2 * kstep, vhi) = x[k2];
            else fc_at<float2>(xb, k1 == 0 ? vlo + (unsigned)(12 * kstep * 4) : vhi - (unsigned)(12 * kstep * 4)) = x[k2];
            if (k2 == 0 || k2 == 12) {
                if (k1 == 0) fc_at<float2>(xb + k2 * kstep, vlo + 128) = x[k2];
            }
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(fmax_) : "v"(x[k2].x), "v"(x[k2].y));
        }
        fmax_ = wave_max(fmax_);
        if ((tid & 63) == 0) atomicMax(reinterpret_cast<int *>(a.amaxf) + b, __float_as_int(fmax_));
    }
}

// ---------------------------------------------------------------------------------------------
// Per-frequency-pair GEMM: Yf[fp][b][h][n] = sum_{dy, s, p, e} Xf[fp][b][h + dy - 1][(s, p, e)] G[fp][(dy, s, p, e)][n]
// 256 threads: wave (wm, wn) takes five of the chunk's ten 16-row M-tiles and two of the four N-tiles; its weight
// fragments (8 k-steps x 2 N-tiles x h / l) stay in registers for the whole workgroup, the activations of a chunk of
// eight windows (160 rows) are split into f16 h / l planes in LDS.
// ---------------------------------------------------------------------------------------------
#define FC_CW 8                         // windows per chunk
#define FC_APITCH 80                    // halfs per staged row (64 + 16 pad = ten 16-byte slots: the four lane groups of a
                                        // ds_read_b128 each touch sixteen different slots; nine slots gave two-way conflicts)

struct FcGemmArgs {
    const float *Xf; float *Yf;
    const float *amaxf;                  // [B]
    const _Float16 *gw;                  // [289][8 k-steps][2 planes][64 n][32 k]
    const int *gsw;                      // [289] weights of pair fp were scaled by 2^gsw
    int B, H, nchunk_per_wg;
    int sf, sh;                          // float strides of the frequency tensors: pair fp, image row h
};

#define FC_OPITCH 36                    // floats per row of a wave's output transposition patch

__global__ __launch_bounds__(256, 2) void fc_gemm_kernel(FcGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fcg_smem[];
    const int H = a.H;
    const int rows_cap = FC_CW * H;
    _Float16 *ah = reinterpret_cast<_Float16 *>(fcg_smem);                   // [rows_cap + 1][FC_APITCH]; the last row is zeros
    _Float16 *al = ah + (size_t)(rows_cap + 1) * FC_APITCH;
    float *patch = reinterpret_cast<float *>(al + (size_t)(rows_cap + 1) * FC_APITCH);       // [4 waves][16][FC_OPITCH]
    float *sa_s = patch + 4 * 16 * FC_OPITCH;                                // [FC_CW] 2^sa, then [FC_CW] 2^-(sa + sw)
    unsigned int *roff = reinterpret_cast<unsigned int *>(sa_s + 2 * FC_CW); // [rows_cap] byte offset of a chunk row in Xf / Yf
    unsigned char *rwin = reinterpret_cast<unsigned char *>(roff + rows_cap);        // [rows_cap] window of the row
    unsigned char *rh = rwin + rows_cap;                                             // [rows_cap] image row
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int wm = wid >> 1, wn = wid & 1;
    const int fp = blockIdx.x;                     // fastest: concurrently running workgroups read neighbouring 5 KB blocks
    // row tables (the same for every chunk) and the zero row the out-of-image taps read
    for (int r = tid; r < rows_cap; r += 256) {
        const int wi = r / H, hr = r - wi * H;
        roff[r] = (unsigned int)(((size_t)wi * FC_NP * H * 64 + (size_t)hr * a.sh) * 4);
        rwin[r] = (unsigned char)wi; rh[r] = (unsigned char)hr;
    }
    if (tid < FC_APITCH) { ah[(size_t)rows_cap * FC_APITCH + tid] = (_Float16)0.f; al[(size_t)rows_cap * FC_APITCH + tid] = (_Float16)0.f; }
    // weight fragments of this wave's two N-tiles, all eight k-steps, both planes
    fc_h8 bh[8][2], bl[8][2];
    {
        const _Float16 *g = a.gw + (size_t)fp * (8 * 2 * 64 * 32);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = 16 * (2 * wn + nt) + (lane & 15);
                bh[ks][nt] = *reinterpret_cast<const fc_h8 *>(g + ((size_t)(ks * 2 + 0) * 64 + n) * 32 + 8 * (lane >> 4));
                bl[ks][nt] = *reinterpret_cast<const fc_h8 *>(g + ((size_t)(ks * 2 + 1) * 64 + n) * 32 + 8 * (lane >> 4));
            }
    }
    const int sw = a.gsw[fp];
    const int nchunks = (a.B + FC_CW - 1) / FC_CW;
    constexpr int NLD = (FC_CW * 20 * 16 + 255) / 256;      // staged float4 per thread (H <= 20; the host refuses taller images)
    fc_f4 pre[NLD];
    __syncthreads();                                        // the row tables are in place
    auto fetch = [&](int chunk) {                           // a chunk's activations -> registers (clamped addresses, no branch)
        const int b0 = chunk * FC_CW;
        const int rows = min(FC_CW, a.B - b0) * H;
        const unsigned char *base = reinterpret_cast<const unsigned char *>(a.Xf + (size_t)b0 * FC_NP * H * 64 + (size_t)fp * a.sf);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = min(tid + 256 * u, rows * 16 - 1);
            pre[u] = fc_at<fc_f4>(base, roff[i >> 4] + (unsigned)((i & 15) * 16));
        }
    };
    int chunk = blockIdx.y * a.nchunk_per_wg;
    if (chunk < nchunks) fetch(chunk);
    for (int it = 0; it < a.nchunk_per_wg; ++it, ++chunk) {
        if (chunk >= nchunks) break;                        // uniform
        const int b0 = chunk * FC_CW;
        const int nw = min(FC_CW, a.B - b0);
        const int rows = nw * H;
        __syncthreads();                                    // the previous chunk's fragments have been read
        if (tid < FC_CW) {
            const int s_ = tid < nw ? fc_scale_exp(a.amaxf[b0 + tid]) : 0;
            sa_s[tid] = ldexpf(1.0f, s_);
            sa_s[FC_CW + tid] = ldexpf(1.0f, -(s_ + sw));
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = tid + 256 * u;
            if (i >= rows * 16) continue;
            const int row = i >> 4, c4 = i & 15;
            const fc_f4 v = pre[u] * sa_s[rwin[row]];
            const fc_h4 hh = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const fc_h4 ll = {(_Float16)((v.x - (float)hh.x) * FC_LSCALE), (_Float16)((v.y - (float)hh.y) * FC_LSCALE),
                              (_Float16)((v.z - (float)hh.z) * FC_LSCALE), (_Float16)((v.w - (float)hh.w) * FC_LSCALE)};
            *reinterpret_cast<fc_h4 *>(ah + (size_t)row * FC_APITCH + 4 * c4) = hh;
            *reinterpret_cast<fc_h4 *>(al + (size_t)row * FC_APITCH + 4 * c4) = ll;
        }
        __syncthreads();
        // the next chunk's activations travel while this one is multiplied
        if (it + 1 < a.nchunk_per_wg && chunk + 1 < nchunks) fetch(chunk + 1);
        const int n_mt = (rows + 15) >> 4;
        unsigned char *ybase = reinterpret_cast<unsigned char *>(a.Yf + (size_t)b0 * FC_NP * H * 64 + (size_t)fp * a.sf);
        float *pt = patch + wid * 16 * FC_OPITCH;
        for (int mt = wm; mt < n_mt; mt += 2) {
            const int r = min(16 * mt + (lane & 15), rows_cap - 1);          // this lane's A row
            const int hrow = rh[r];
            const bool rin = 16 * mt + (lane & 15) < rows;
            fc_f4 hi[2], lo[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { hi[nt] = fc_f4{0.f, 0.f, 0.f, 0.f}; lo[nt] = fc_f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const int hs = hrow + dy - 1;
                const int rs = (rin && hs >= 0 && hs < H) ? r + dy - 1 : rows_cap;       // the zero row outside the image
                const _Float16 *pa = ah + (size_t)rs * FC_APITCH + 8 * (lane >> 4), *pl = al + (size_t)rs * FC_APITCH + 8 * (lane >> 4);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const int ks = 2 * dy + s_;
                    const fc_h8 fa = *reinterpret_cast<const fc_h8 *>(pa + 32 * s_);
                    const fc_h8 fl = *reinterpret_cast<const fc_h8 *>(pl + 32 * s_);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        hi[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, bh[ks][nt], hi[nt], 0, 0, 0);
                        lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, bl[ks][nt], lo[nt], 0, 0, 0);
                        lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl, bh[ks][nt], lo[nt], 0, 0, 0);
                    }
                }
            }
            // D (column n = lane & 15, rows 4 (lane >> 4) + e) goes through the wave's LDS patch so that every lane stores
            // 16 bytes and a row's 32 columns leave as one 128-byte line
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pt[(4 * (lane >> 4) + e) * FC_OPITCH + 16 * nt + (lane & 15)] = hi[nt][e] + lo[nt][e] * (1.0f / FC_LSCALE);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rr = 8 * half + (lane >> 3), ro = 16 * mt + rr;
                fc_f4 v = *reinterpret_cast<const fc_f4 *>(pt + rr * FC_OPITCH + 4 * (lane & 7));
                if (ro < rows) {
                    v *= sa_s[FC_CW + rwin[ro]];
                    fc_at<fc_f4>(ybase, roff[ro] + (unsigned)((32 * wn + 4 * (lane & 7)) * 4)) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// =====================================================================================
// Host side
// =====================================================================================
struct amt_fftconv_layer {
    _Float16 *gw = nullptr;
    int *gsw = nullptr;
    float2 *tw = nullptr;
    float *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;      // optional device copies (stand-alone entry)
};

static inline unsigned short fc_f16_bits(_Float16 h) { unsigned short b; memcpy(&b, &h, 2); return b; }

int amt_fftconv_layer_create_internal(amt_fftconv_layer **out, const float *kernel /* [4][16][32][32] host */) {
    if (!out || !kernel) return AMT_E_INVALID;
    const int KH = 4, KW = 16, C = 32, P = 16, pl = (KW - 1) / 2;
    amt_fftconv_layer *L = new amt_fftconv_layer();
    std::vector<_Float16> gw((size_t)FC_NP * 8 * 2 * 64 * 32);
    std::vector<int> gsw(FC_NP);
    std::vector<double> G(256 * 64);
    const double PI2 = 6.283185307179586476925286766559;
    // Kf[fo][dy][ci][co] = sum_dx K[dy][dx][ci][co] e^{+2 pi i fo (dx - pl) / 576} for every fo (a [576 x 16] x [16 x 4096]
    // complex product), so that Y[w] = IDFT(Af Kf)[w] is the 'same'-padded correlation of the layer
    std::vector<double> cs(FC_NF), sn(FC_NF);
    for (int m = 0; m < FC_NF; ++m) { cs[m] = cos(PI2 * m / FC_NF); sn[m] = sin(PI2 * m / FC_NF); }
    const size_t KSZ = (size_t)KH * C * C;
    std::vector<double> kfr((size_t)FC_NF * KSZ, 0.0), kfi((size_t)FC_NF * KSZ, 0.0);
    for (int fo = 0; fo < FC_NF; ++fo)
        for (int dx = 0; dx < KW; ++dx) {
            const int m = (int)((((long)fo * (dx - pl)) % FC_NF + FC_NF) % FC_NF);
            const double c_ = cs[m], s_ = sn[m];
            for (int dy = 0; dy < KH; ++dy) {
                const float *kp = kernel + (size_t)(dy * KW + dx) * C * C;
                double *orr = kfr.data() + (size_t)fo * KSZ + (size_t)dy * C * C, *oii = kfi.data() + (size_t)fo * KSZ + (size_t)dy * C * C;
                for (int e = 0; e < C * C; ++e) { orr[e] += kp[e] * c_; oii[e] += kp[e] * s_; }
            }
        }
    for (int fp = 0; fp < FC_NP; ++fp) {
        std::fill(G.begin(), G.end(), 0.0);
        for (int t = 0; t < 2; ++t) {
            const int fo = t == 0 ? fp : (FC_NF - fp) % FC_NF;
            const double *kre = kfr.data() + (size_t)fo * KSZ, *kim = kfi.data() + (size_t)fo * KSZ;
            const int s_z = t == 0 ? 0 : 1, s_c = t == 0 ? 1 : 0;
            for (int dy = 0; dy < KH; ++dy)
                for (int p = 0; p < P; ++p)
                    for (int q = 0; q < P; ++q) {
                        // cz[c] = (Kf[2p][c] - i Kf[2p+1][c]) / 2, cc[c] = (Kf[2p][c] + i Kf[2p+1][c]) / 2, c in {2q, 2q+1};
                        // wz = cz[2q] + i cz[2q+1], wc = cc[2q] + i cc[2q+1]
                        auto K = [&](int ci, int co, double &re, double &im) {
                            re = kre[((size_t)dy * C + ci) * C + co]; im = kim[((size_t)dy * C + ci) * C + co];
                        };
                        double er0, ei0, er1, ei1, or0, oi0, or1, oi1;
                        K(2 * p, 2 * q, er0, ei0); K(2 * p, 2 * q + 1, er1, ei1);
                        K(2 * p + 1, 2 * q, or0, oi0); K(2 * p + 1, 2 * q + 1, or1, oi1);
                        // cz[c] = (e - i o) / 2 = ((er + oi) + i (ei - or)) / 2;  cc[c] = (e + i o) / 2 = ((er - oi) + i (ei + or)) / 2
                        const double cz0r = 0.5 * (er0 + oi0), cz0i = 0.5 * (ei0 - or0), cz1r = 0.5 * (er1 + oi1), cz1i = 0.5 * (ei1 - or1);
                        const double cc0r = 0.5 * (er0 - oi0), cc0i = 0.5 * (ei0 + or0), cc1r = 0.5 * (er1 - oi1), cc1i = 0.5 * (ei1 + or1);
                        // w = c0 + i c1
                        const double wzr = cz0r - cz1i, wzi = cz0i + cz1r, wcr = cc0r - cc1i, wci = cc0i + cc1r;
                        const int kz = ((dy * 2 + s_z) * P + p) * 2, kc = ((dy * 2 + s_c) * P + p) * 2, n = (t * P + q) * 2;
                        G[(size_t)kz * 64 + n] += wzr; G[(size_t)(kz + 1) * 64 + n] += -wzi;
                        G[(size_t)kz * 64 + n + 1] += wzi; G[(size_t)(kz + 1) * 64 + n + 1] += wzr;
                        G[(size_t)kc * 64 + n] += wcr; G[(size_t)(kc + 1) * 64 + n] += wci;
                        G[(size_t)kc * 64 + n + 1] += wci; G[(size_t)(kc + 1) * 64 + n + 1] += -wcr;
                    }
        }
        double gmax = 0;
        for (double v : G) gmax = std::max(gmax, std::fabs(v));
        int sw = 0;
        if (gmax > 0) { int e; (void)frexp(gmax, &e); sw = 4 - e; }       // max |G| 2^sw in [8, 16)
        gsw[fp] = sw;
        const double sc = ldexp(1.0, sw);
        for (int k = 0; k < 256; ++k)
            for (int n = 0; n < 64; ++n) {
                const float v = (float)(G[(size_t)k * 64 + n] * sc);
                _Float16 hh = (_Float16)v;
                if (!(std::fabs(v) >= 6.103515625e-05f)) hh = (_Float16)0.0f;
                const float rr = (v - (float)hh) * FC_LSCALE;
                _Float16 ll = (_Float16)rr;
                if (!(std::fabs(rr) >= 6.103515625e-05f)) ll = (_Float16)0.0f;
                const int ks = k >> 5, kk = k & 31;
                gw[(((size_t)fp * 8 + ks) * 2 + 0) * 64 * 32 + (size_t)n * 32 + kk] = hh;
                gw[(((size_t)fp * 8 + ks) * 2 + 1) * 64 * 32 + (size_t)n * 32 + kk] = ll;
            }
    }
    std::vector<float2> tw(FC_NF);
    for (int m = 0; m < FC_NF; ++m) {
        const double ang = -PI2 * m / FC_NF;
        tw[m] = make_float2((float)cos(ang), (float)sin(ang));
    }
    hipError_t e = hipMalloc(&L->gw, gw.size() * sizeof(_Float16));
    if (e == hipSuccess) e = hipMemcpy(L->gw, gw.data(), gw.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&L->gsw, gsw.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(L->gsw, gsw.data(), gsw.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&L->tw, tw.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(L->tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        snprintf(amt_hip_err_buf, sizeof(amt_hip_err_buf), "fftconv upload: %s", hipGetErrorString(e));
        amt_fftconv_layer_destroy_internal(L);
        return AMT_E_HIP;
    }
    *out = L;
    return AMT_OK;
}

void amt_fftconv_layer_destroy_internal(amt_fftconv_layer *L) {
    if (!L) return;
    for (void *p : {(void *)L->gw, (void *)L->gsw, (void *)L->tw, (void *)L->s1, (void *)L->t1, (void *)L->s2, (void *)L->t2})
        if (p) (void)hipFree(p);
    delete L;
}

size_t amt_fftconv_freq_floats(int B, int H) { return (size_t)FC_NP * B * H * 64; }

// a window's block is [289 pairs][H rows][64]: a GEMM workgroup reads H x 256 contiguous bytes per pair (the row-major
// alternative, 74 KB contiguous per row transform and 256-byte pieces for the GEMM, measured 2 % slower in round 3)
static void fc_strides(int H, int *sf, int *sh) { *sf = H * 64; *sh = 64; }

static const size_t FC_ROW_LDS = (size_t)(16 * FC_PS) * 4 + FC_NF * sizeof(float2);
template <bool IN_FREQ, int EPI>
static int fc_row_launch_t(const FcRowArgs &a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)fc_row_kernel<IN_FREQ, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FC_ROW_LDS));
        attr = true;
    }
    fc_row_kernel<IN_FREQ, EPI><<<a.B * a.H, FC_THREADS, FC_ROW_LDS, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
template <bool IN_FREQ>
static int fc_row_launch(const FcRowArgs &a, hipStream_t st) {
    if (!IN_FREQ) return fc_row_launch_t<false, 1>(a, st);
    if (!a.out_sp) {
        if (a.sc || a.sc1 || a.amax_out || !a.Xf) return AMT_E_UNSUPPORTED;        // (no layer of the chain: a shortcut sum is always read later)
        return fc_row_launch_t<true, 1>(a, st);
    }
    if (a.sc) return fc_row_launch_t<true, 2>(a, st);
    if (a.sc1) return fc_row_launch_t<true, 3>(a, st);
    return fc_row_launch_t<true, 4>(a, st);
}

int amt_fftconv_forward_fft(const amt_fftconv_layer *L, const float *in_sp, size_t in_stride, int B, int H, int W,
                            float *Xf, float *amaxf, hipStream_t st) {
    if (!L || !in_sp || !Xf || !amaxf || W + 15 > FC_NF) return AMT_E_INVALID;
    AMT_HIP_CHECK(hipMemsetAsync(amaxf, 0, (size_t)B * sizeof(float), st));
    FcRowArgs a{};
    a.in_sp = in_sp; a.in_stride = in_stride; a.Xf = Xf; a.amaxf = amaxf; a.tw = L->tw; a.B = B; a.H = H; a.W = W;
    fc_strides(H, &a.sf, &a.sh);
    return fc_row_launch<false>(a, st);
}

int amt_fftconv_gemm(const amt_fftconv_layer *L, const float *Xf, const float *amaxf, int B, int H, float *Yf, hipStream_t st) {
    if (!L || !Xf || !Yf || !amaxf) return AMT_E_INVALID;
    if (H > 20) return AMT_E_UNSUPPORTED;
    const size_t rows_cap = (size_t)FC_CW * H;
    const size_t lds = 2 * (rows_cap + 1) * FC_APITCH * sizeof(_Float16) + (size_t)4 * 16 * FC_OPITCH * 4 + 2 * FC_CW * 4 + rows_cap * 4 +
                       2 * rows_cap + 16;
    static size_t attr = 0;
    if (lds > attr) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)fc_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    const int nchunks = (B + FC_CW - 1) / FC_CW;
    int per = 1;
    while ((size_t)((nchunks + per - 1) / per) * FC_NP > 4096 && per < 16) per *= 2;      // ~2 resident rounds of workgroups
    FcGemmArgs a{Xf, Yf, amaxf, L->gw, L->gsw, B, H, per, 0, 0};
    fc_strides(H, &a.sf, &a.sh);
    fc_gemm_kernel<<<dim3(FC_NP, (nchunks + per - 1) / per), 256, lds, st>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_fftconv_inverse_epilogue(const amt_fftconv_layer *L, const float *Yf, const FcEpilogue &ep, int B, int H, int W,
                                 float *out_sp, size_t out_stride, float *Xf_next, float *amaxf_next, float *amax_out,
                                 hipStream_t st) {
    if (!L || !Yf || !ep.s1 || !ep.t1 || (!out_sp && !Xf_next)) return AMT_E_INVALID;
    if (Xf_next && !amaxf_next) return AMT_E_INVALID;
    if (Xf_next) AMT_HIP_CHECK(hipMemsetAsync(amaxf_next, 0, (size_t)B * sizeof(float), st));
    FcRowArgs a{};
    a.Yf = Yf; a.out_sp = out_sp; a.out_stride = out_stride; a.Xf = Xf_next; a.amaxf = amaxf_next; a.amax_out = amax_out;
    a.s1 = ep.s1; a.t1 = ep.t1; a.s2 = ep.s2; a.t2 = ep.t2; a.sc = ep.sc; a.sc_stride = ep.sc_stride;
    a.sc1 = ep.sc1; a.sc1_stride = ep.sc1_stride; a.sc1_w = ep.sc1_w; a.sc1_s = ep.sc1_s; a.sc1_t = ep.sc1_t;
    a.tw = L->tw; a.B = B; a.H = H; a.W = W;
    fc_strides(H, &a.sf, &a.sh);
    return fc_row_launch<true>(a, st);
}

// ---- stand-alone C ABI entry (tests / microbenchmarks): one layer, spatial in, spatial out --------------------------
extern "C" {

int amt_fftconv_create(amt_fftconv_layer **layer, const float *kernel_host, const float *s1, const float *t1,
                       const float *s2, const float *t2) {
    if (!layer || !kernel_host || !s1 || !t1) return AMT_E_INVALID;
    int rc = amt_fftconv_layer_create_internal(layer, kernel_host);
    if (rc != AMT_OK) return rc;
    amt_fftconv_layer *L = *layer;
    auto up = [&](const float *h, float **d) -> hipError_t {
        if (!h) return hipSuccess;
        hipError_t e = hipMalloc(d, 32 * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(*d, h, 32 * sizeof(float), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(s1, &L->s1);
    if (e == hipSuccess) e = up(t1, &L->t1);
    if (e == hipSuccess) e = up(s2, &L->s2);
    if (e == hipSuccess) e = up(t2, &L->t2);
    if (e != hipSuccess) { amt_fftconv_layer_destroy_internal(L); *layer = nullptr; return AMT_E_HIP; }
    return AMT_OK;
}

int amt_fftconv_destroy(amt_fftconv_layer *layer) { amt_fftconv_layer_destroy_internal(layer); return AMT_OK; }

size_t amt_fftconv_workspace_bytes(int B, int H) { return (2 * amt_fftconv_freq_floats(B, H) + 2 * (size_t)B) * sizeof(float); }

int amt_fftconv_run(const amt_fftconv_layer *L, const float *in, const float *shortcut, int B, int H, int W, float *out,
                    void *workspace, size_t workspace_bytes, int repeat_gemm, void *stream) {
    if (!L || !in || !out || !workspace || B <= 0 || H <= 0 || W <= 0) return AMT_E_INVALID;
    if (W + 15 > FC_NF) return AMT_E_UNSUPPORTED;
    if (workspace_bytes < amt_fftconv_workspace_bytes(B, H)) return AMT_E_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    float *Xf = (float *)workspace, *Yf = Xf + amt_fftconv_freq_floats(B, H), *amaxf = Yf + amt_fftconv_freq_floats(B, H);
    const size_t stride = (size_t)H * W * 32;
    int rc = amt_fftconv_forward_fft(L, in, stride, B, H, W, Xf, amaxf, st);
    for (int i = 0; i < (repeat_gemm > 0 ? repeat_gemm : 1) && rc == AMT_OK; ++i) rc = amt_fftconv_gemm(L, Xf, amaxf, B, H, Yf, st);
    if (rc != AMT_OK) return rc;
    FcEpilogue ep{};
    ep.s1 = L->s1; ep.t1 = L->t1;
    if (shortcut && L->s2) { ep.s2 = L->s2; ep.t2 = L->t2; ep.sc = shortcut; ep.sc_stride = stride; }
    return amt_fftconv_inverse_epilogue(L, Yf, ep, B, H, W, out, stride, nullptr, nullptr, nullptr, st);
}

}  // extern "C"
