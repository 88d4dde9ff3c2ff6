// RDCNN forward (res_net.predict) for gfx950: implicit-GEMM convolutions on the matrix pipe
// with fused BN + sigmoid (+ shortcut add + BN) epilogues, in three f32-equivalent arithmetics:
//   mode 2 (default) split-fp16, amt_conv_f16x3.h;  mode 1 split-bf16, conv_bf16x6_kernel below;
//   mode 0 f32 MFMA, conv_mfma_kernel below (described next).
//
// Replaces keras Model.predict for the graph built in
// /root/reference/RDCNN.py:176-233 (+ _add_shortcut :312-335, output scaling
// :304-310, :591-597) -- see oracle/rdcnn.py for the CPU restatement.
//
// Data layout in HBM: activations NHWC f32, [B][H][W][C]; the flatten order of
// Keras (H, W, C) is then the memory order, so Dense consumes it as is.
//
// Conv kernel (Cin multiple of 32): one 256-thread workgroup computes
// 128*MT output positions x Cout channels.  GEMM view: M = positions,
// N = Cout, K = (dy, dx, cin).  v_mfma_f32_32x32x2_f32 (f32 in, f32 acc: a
// k-ordered fmaf chain per (tap, 32-channel chunk), the chunks' sums added in
// tap order) -- the contraction the north star puts on MFMA while keeping
// float parity with the CPU.
//   * the input tile incl. the conv halo (explicit zeros = Keras "same"
//     padding, asymmetric for even kernels) is staged once per 32-channel
//     chunk into LDS as [pos][33] (pad 1 float: A-fragment reads hit 32
//     distinct banks);
//   * weights are pre-arranged on the host as [cchunk][tap][c][j][nt] so a
//     (tap, chunk) slab is one linear copy into a double-buffered LDS slab
//     and a lane's B fragments for all N-tiles are one ds_read_b32/b64/b128;
//   * one barrier per tap; 16 k-steps x MT x NT MFMAs between barriers;
//   * epilogue: acc*s1+t1 -> sigmoid -> (+shortcut)*s2+t2 -> coalesced NHWC
//     stores (a store instruction = two full 128-B channel rows).
#include "amt_common.h"
#include "amt_fftconv.h"
#include "amt_convh.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RD_CC 32           // channels per staged chunk
#define RD_CSTRIDE 33      // LDS floats per staged position
#define RD_BN_EPS 1e-3f

struct ConvParams {
    const float *in;  size_t in_win_stride;      // [B][H][W][CIN]
    float *out;       size_t out_win_stride;     // [B][H][W][COUT]
    const float *sc;  size_t sc_win_stride;      // shortcut tensor (same shape as out) or null
    const float *w;                              // pre-arranged weights
    const float *s1, *t1, *s2, *t2;              // folded BN (s2/t2 null if no residual)
    int B, H, W;
    int TH, TW, NWIN;                            // workgroup tile
    int tiles_h, tiles_w;
    int cout_total;                              // channels of the output tensor (>= the kernel's COUT
                                                 // when the layer is split over blockIdx.y N-slices)
    // rank-1 shortcut (conv_f16x3s_kernel): the first residual block projects the ONE-channel network
    // input with a 1x1 kernel + BN (RDCNN.py:328-334); instead of materialising that [H][W][COUT]
    // tensor the epilogue forms (x * w_c) * s_c + t_c itself -- the operations of proj_kernel
    const float *sc1 = nullptr; size_t sc1_win_stride = 0;      // [B][H][W] (null: not used)
    const float *sc1_w = nullptr, *sc1_s = nullptr, *sc1_t = nullptr;   // [COUT]
    int pad_t = 0, pad_l = 0;                    // TRAIN form of conv_f16x3s_kernel only: rows / columns of padding before the image
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int KH, int KW, int CIN, int COUT, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    constexpr int NT = COUT / 32;
    constexpr int NCHUNK = CIN / RD_CC;
    constexpr int NTAPS = KH * KW;
    constexpr int PCAP = 128 * MT;               // positions per workgroup
    constexpr int PAD_T = (KH - 1) / 2, PAD_L = (KW - 1) / 2;
    constexpr int WSLAB = RD_CC * COUT;           // floats per (tap, chunk) weight slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf = smem;                           // [2][WSLAB]
    int *pos_sp = reinterpret_cast<int *>(smem + 2 * WSLAB);   // [PCAP] spatial index or -1
    int *pos_win = pos_sp + PCAP;                 // [PCAP] global window
    float *in_lds = reinterpret_cast<float *>(pos_win + PCAP); // [POSIN][33]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int THin = p.TH + KH - 1, TWin = p.TW + KW - 1;
    // block -> (window group, tile row, tile col)
    int bid = blockIdx.x;
    const int tc = bid % p.tiles_w; bid /= p.tiles_w;
    const int tr = bid % p.tiles_h; bid /= p.tiles_h;
    const int win0 = bid * p.NWIN;
    const int r0 = tr * p.TH, c0 = tc * p.TW;
    const int ptile = p.TH * p.TW;

    for (int q = tid; q < PCAP; q += 256) {
        const int w_ = q / ptile, rem = q - w_ * ptile;
        const int r = rem / p.TW, c = rem - r * p.TW;
        const bool ok = w_ < p.NWIN && (win0 + w_) < p.B && (r0 + r) < p.H && (c0 + c) < p.W;
        pos_sp[q] = ok ? (r0 + r) * p.W + (c0 + c) : -1;
        pos_win[q] = win0 + w_;
    }
    // per-lane LDS base (floats) of the A fragment for each of this wave's M-tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int q = (wid * MT + mt) * 32 + (lane & 31);
        int w_ = q / ptile, rem = q - w_ * ptile;
        int r = rem / p.TW, c = rem - r * p.TW;
        if (w_ >= p.NWIN) { w_ = 0; r = 0; c = 0; }       // padding rows of the tile: any valid address
        abase[mt] = ((w_ * THin + r) * TWin + c) * RD_CSTRIDE + (lane >> 5);
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

    constexpr int WV4 = WSLAB / 4 / 256;          // float4 per thread per slab (1, 2 or 4)
    static_assert(WSLAB % (4 * 256) == 0, "slab split");
    // N-slice of this workgroup (layers whose output-tile grid cannot fill the chip are split
    // over blockIdx.y into COUT-wide channel slices, each with its own weight block)
    const int cout_off = blockIdx.y * COUT;
    const float4 *wg4 = reinterpret_cast<const float4 *>(p.w + (size_t)blockIdx.y * ((size_t)CIN * NTAPS * COUT));

    for (int ch = 0; ch < NCHUNK; ++ch) {
        __syncthreads();                          // previous chunk fully consumed
        // ---- stage the input tile (32 channels) with its zero halo -------------
        {
            const int c4 = tid & 7;               // float4 within the 32 channels
            const int cl = tid >> 3;              // column lane 0..31
            for (int wr = 0; wr < p.NWIN * THin; ++wr) {
                const int w_ = wr / THin, ri = wr - w_ * THin;
                const int gr = r0 - PAD_T + ri;
                const int gw = win0 + w_;
                const bool rok = gr >= 0 && gr < p.H && gw < p.B;
                const float *src = p.in + (size_t)gw * p.in_win_stride +
                                   ((size_t)gr * p.W) * CIN + ch * RD_CC + c4 * 4;
                float *dst = in_lds + (size_t)wr * TWin * RD_CSTRIDE + c4 * 4;
                for (int ci = cl; ci < TWin; ci += 32) {
                    const int gc = c0 - PAD_L + ci;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rok && gc >= 0 && gc < p.W)
                        v = *reinterpret_cast<const float4 *>(src + (size_t)gc * CIN);
                    float *d = dst + ci * RD_CSTRIDE;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            }
        }
        // ---- first weight slab of this chunk ---------------------------------
        {
            const float4 *src = wg4 + (size_t)(ch * NTAPS) * (WSLAB / 4);
            float4 *dst = reinterpret_cast<float4 *>(wbuf);
#pragma unroll
            for (int i = 0; i < WV4; ++i) dst[tid + i * 256] = src[tid + i * 256];
        }
        __syncthreads();
        int cur = 0;
        for (int dy = 0; dy < KH; ++dy) {
            for (int dx = 0; dx < KW; ++dx) {
                const int tap = dy * KW + dx;
                float4 wpre[WV4];
                const bool more = tap + 1 < NTAPS;
                if (more) {
                    const float4 *src = wg4 + (size_t)(ch * NTAPS + tap + 1) * (WSLAB / 4);
#pragma unroll
                    for (int i = 0; i < WV4; ++i) wpre[i] = src[tid + i * 256];
                }
                const int tapoff = (dy * TWin + dx) * RD_CSTRIDE;
                const float *wb = wbuf + cur * WSLAB + ((lane >> 5) * 32 + (lane & 31)) * NT;
                // blocked summation: the 32 products of one (tap, chunk) go through a fresh accumulator (an fmaf
                // chain of length 32 inside the matrix pipe) which is then added to the running sum -- 64 + 32
                // roundings on the critical path of a K = 2048 contraction instead of 2048.  A single chain over
                // all of K (round 1-2) sat 11x farther from the float64 result than numpy's blocked GEMM on the
                // timing head at N = 2048 (profiles/r02/rdcnn_error_vs_f64.json); OpenBLAS blocks K the same way.
                f32x16 part[MT][NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) part[mt][nt][e] = 0.f;
#pragma unroll
                for (int cp = 0; cp < RD_CC / 2; ++cp) {
                    float a[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = in_lds[abase[mt] + tapoff + 2 * cp];
                    float bfr[NT];
                    const float *wrow = wb + (2 * cp) * 32 * NT;
                    if constexpr (NT == 1) {
                        bfr[0] = wrow[0];
                    } else if constexpr (NT == 2) {
                        const float2 t2 = *reinterpret_cast<const float2 *>(wrow);
                        bfr[0] = t2.x; bfr[1] = t2.y;
                    } else {
                        const float4 t4 = *reinterpret_cast<const float4 *>(wrow);
                        bfr[0] = t4.x; bfr[1] = t4.y; bfr[2] = t4.z; bfr[3] = t4.w;
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            part[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a[mt], bfr[nt], part[mt][nt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[mt][nt][e] = __fadd_rn(acc[mt][nt][e], part[mt][nt][e]);
                if (more) {
                    float4 *dst = reinterpret_cast<float4 *>(wbuf + (cur ^ 1) * WSLAB);
#pragma unroll
                    for (int i = 0; i < WV4; ++i) dst[tid + i * 256] = wpre[i];
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    // ---- epilogue ---------------------------------------------------------------
    const int j = lane & 31;
    const int CT = p.cout_total;
    float s1[NT], t1[NT], s2[NT], t2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = p.s1[cout_off + nt * 32 + j]; t1[nt] = p.t1[cout_off + nt * 32 + j];
        s2[nt] = p.s2 ? p.s2[cout_off + nt * 32 + j] : 1.f;
        t2[nt] = p.t2 ? p.t2[cout_off + nt * 32 + j] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int q = (wid * MT + mt) * 32 + row;
            const int sp = pos_sp[q];
            if (sp < 0) continue;
            const int gw = pos_win[q];
            float *o = p.out + (size_t)gw * p.out_win_stride + (size_t)sp * CT + cout_off + j;
            const float *scp = p.sc ? p.sc + (size_t)gw * p.sc_win_stride + (size_t)sp * CT + cout_off + j : nullptr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v = sigmoidf_(acc[mt][nt][e] * s1[nt] + t1[nt]);
                if (scp) v = (v + scp[nt * 32]) * s2[nt] + t2[nt];
                o[nt * 32] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Split-bf16 convolution ("bf16x6"): fp32-equivalent products on the bf16 MFMA pipe.
// Every f32 operand x is split exactly into three bf16 terms x = x1 + x2 + x3
// (8 + 8 + 8 mantissa bits); the product x*y is the six terms with i + j <= 4
// (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1), each exact in the MFMA's f32 accumulate; the
// dropped terms are <= 2^-26 |xy|, below f32 rounding.  Six v_mfma_f32_32x32x16_bf16
// per 16-deep k-block replace eight v_mfma_f32_32x32x2_f32: 2.67x fewer matrix-pipe
// cycles at f32 accuracy (bf16 MFMA = 16x the f32 MFMA rate, MI355X_MICROARCH.md).
//   * 512-thread workgroup, 8 waves x one 32-position M-tile; same tile geometry,
//     halo staging and epilogue as the f32 kernel;
//   * activations stay f32 in HBM and are split once per workgroup while staging
//     into LDS as [pos][plane(3)][32 ch] bf16 (208-B pitch: conflict-free
//     ds_read_b128 A fragments, lane = position, 8 consecutive channels);
//   * weights are split on the host and laid out so a (tap, k-block, plane, N-tile)
//     fragment is one conflict-free ds_read_b128 per lane.
// ---------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define BX_CC 16                             // channels per staged chunk (one bf16 k-block)
#define BX_PSTRIDE 112                      // bytes per staged position: 3 planes x 16 ch x 2 B + 16 pad
__host__ __device__ __forceinline__ int bx_row_pitch(int TW, int TWin) {
    return TWin + (((TW - TWin) % 16) + 16) % 16;
}

__host__ __device__ __forceinline__ unsigned short amt_f2bf(float x) {
    unsigned int u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(x);
#else
    memcpy(&u, &x, 4);
#endif
    u += 0x7FFFu + ((u >> 16) & 1u);         // round to nearest even
    return (unsigned short)(u >> 16);
}
__host__ __device__ __forceinline__ float amt_bf2f(unsigned short h) {
    unsigned int u = ((unsigned int)h) << 16;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
__host__ __device__ __forceinline__ void amt_split3(float x, unsigned short &h1, unsigned short &h2,
                                                    unsigned short &h3) {
    h1 = amt_f2bf(x);
    const float r1 = x - amt_bf2f(h1);
    h2 = amt_f2bf(r1);
    const float r2 = r1 - amt_bf2f(h2);
    h3 = amt_f2bf(r2);
}

// MASKED = true is the small-image form (whole H x W image of NWIN windows per workgroup):
// the LDS tile holds only real positions plus one all-zero position, and every tap's A
// fragment address is chosen per lane (in-bounds neighbour or the zero position), so the
// "same" padding costs no LDS -- a 5x8 image with a 4x16 kernel would otherwise stage 4.6x its
// size in halo zeros.  Layers are additionally split over blockIdx.y into COUT-wide slices.
template <int KH, int KW, int CIN, int COUT, bool MASKED>
__global__ __launch_bounds__(512, 4) void conv_bf16x6_kernel(ConvParams p, const uint4 *__restrict__ w16s) {
    // Geometry: 8 waves x one 32-position M-tile = 256 output positions per workgroup; the
    // contraction is walked in 16-channel chunks (one bf16 k-block per tap), so the staged
    // input tile is [pos][plane(3)][16 ch] bf16 = 112 B per position and TWO workgroups fit
    // a CU: one stages / stores while the other keeps the matrix pipe busy.
    constexpr int NT = COUT / 32;
    constexpr int NCHUNK = CIN / BX_CC;
    constexpr int NTAPS = KH * KW;
    constexpr int PCAP = 256;
    constexpr int PAD_T = (KH - 1) / 2, PAD_L = (KW - 1) / 2;
    constexpr int TPS = (NT == 1 && NTAPS % 2 == 0) ? 2 : 1;       // taps per weight slab
    constexpr int NSLAB = NTAPS / TPS;
    constexpr int SLAB_V4 = TPS * 3 * NT * 64;                      // uint4 per slab (fragment = 64 lanes x 16 B)
    constexpr int WV4 = (SLAB_V4 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 *wbuf = reinterpret_cast<uint4 *>(smem);                  // [3][SLAB_V4]
    int *pos_sp = reinterpret_cast<int *>(wbuf + 3 * SLAB_V4);      // [PCAP]
    int *pos_win = pos_sp + PCAP;
    char *in_lds = reinterpret_cast<char *>(pos_win + PCAP);        // [POSIN][112 B]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int THin = MASKED ? p.TH : p.TH + KH - 1, TWin = MASKED ? p.TW : p.TW + KW - 1;
    // LDS row pitch (in positions) == TW (mod 16): consecutive flattened tile positions stay an
    // odd number (7) of 16-byte slots apart across the row wrap -> conflict-free b128 reads
    const int RP = MASKED ? p.TW : bx_row_pitch(p.TW, TWin);
    const int cout_off = blockIdx.y * COUT;                         // N-slice of this workgroup
    const uint4 *w16 = w16s + (size_t)blockIdx.y * ((size_t)NCHUNK * NSLAB * SLAB_V4);
    int bid = blockIdx.x;
    const int tc = bid % p.tiles_w; bid /= p.tiles_w;
    const int tr = bid % p.tiles_h; bid /= p.tiles_h;
    const int win0 = bid * p.NWIN;
    const int r0 = tr * p.TH, c0 = tc * p.TW;
    const int ptile = p.TH * p.TW;

    for (int q = tid; q < PCAP; q += 512) {
        const int w_ = q / ptile, rem = q - w_ * ptile;
        const int r = rem / p.TW, c = rem - r * p.TW;
        const bool ok = w_ < p.NWIN && (win0 + w_) < p.B && (r0 + r) < p.H && (c0 + c) < p.W;
        pos_sp[q] = ok ? (r0 + r) * p.W + (c0 + c) : -1;
        pos_win[q] = win0 + w_;
    }
    int abase, lr, lc;                                              // lane's LDS base and tile coords
    {
        int q = wid * 32 + (lane & 31);
        int w_ = q / ptile, rem = q - w_ * ptile;
        int r = rem / p.TW, c = rem - r * p.TW;
        if (w_ >= p.NWIN) { w_ = 0; r = 0; c = 0; }
        abase = ((w_ * THin + r) * RP + c) * BX_PSTRIDE + (lane >> 5) * 16;
        lr = r; lc = c;
    }
    const int zero_off = p.NWIN * THin * RP * BX_PSTRIDE + (lane >> 5) * 16;   // MASKED: the zero position
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    for (int ch = 0; ch < NCHUNK; ++ch) {
        __syncthreads();
        // ---- stage + split the input tile (16 channels) --------------------------------
        // work item = (halo row, column, 8-channel half); items are dealt round-robin to the
        // 512 threads and ALL of a thread's global loads are issued before the first split,
        // so the tile costs one memory latency instead of one per halo row.
        {
            const int nrow = p.NWIN * THin;
            const int items = nrow * TWin * 2;
            constexpr int MAXIT = 2;
            for (int it0 = 0; it0 < items; it0 += 512 * MAXIT) {
                float4 v0[MAXIT], v1[MAXIT];
                int dsto[MAXIT];
#pragma unroll
                for (int u = 0; u < MAXIT; ++u) {
                    const int it = it0 + u * 512 + tid;
                    v0[u] = make_float4(0.f, 0.f, 0.f, 0.f); v1[u] = v0[u]; dsto[u] = -1;
                    if (it < items) {
                        const int cg = it & 1;
                        const int pc = it >> 1;
                        const int wr = pc / TWin, ci = pc - wr * TWin;
                        const int w_ = wr / THin, ri = wr - w_ * THin;
                        const int gr = r0 + ri - (MASKED ? 0 : PAD_T), gc = c0 + ci - (MASKED ? 0 : PAD_L), gw = win0 + w_;
                        dsto[u] = (wr * RP + ci) * BX_PSTRIDE + cg * 16;
                        if (gr >= 0 && gr < p.H && gw < p.B && gc >= 0 && gc < p.W) {
                            const float4 *src = reinterpret_cast<const float4 *>(
                                p.in + (size_t)gw * p.in_win_stride + ((size_t)gr * p.W + gc) * CIN + ch * BX_CC + cg * 8);
                            v0[u] = src[0]; v1[u] = src[1];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < MAXIT; ++u) {
                    if (dsto[u] < 0) continue;
                    const float v[8] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w, v1[u].x, v1[u].y, v1[u].z, v1[u].w};
                    unsigned short h[3][8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) amt_split3(v[e], h[0][e], h[1][e], h[2][e]);
                    char *dst = in_lds + dsto[u];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        uint4 pk;
                        pk.x = h[pl][0] | ((unsigned)h[pl][1] << 16);
                        pk.y = h[pl][2] | ((unsigned)h[pl][3] << 16);
                        pk.z = h[pl][4] | ((unsigned)h[pl][5] << 16);
                        pk.w = h[pl][6] | ((unsigned)h[pl][7] << 16);
                        *reinterpret_cast<uint4 *>(dst + pl * 32) = pk;
                    }
                }
            }
        }
        if (MASKED && tid < 7) {                                     // the all-zero position (112 B)
            *reinterpret_cast<uint4 *>(in_lds + p.NWIN * THin * RP * BX_PSTRIDE + tid * 16) = make_uint4(0, 0, 0, 0);
        }
        // ---- K loop over weight slabs: 3 LDS buffers, global prefetch two slabs ahead ------
        //  step s: issue the loads of slab s+2 (registers), run slab s from LDS, then park
        //  slab s+1 (loaded during step s-1: two slab-times of latency budget) into LDS.
        //  The loads are inline asm on purpose: hipcc sinks an ordinary prefetch load down to
        //  its use and waits vmcnt(0) there (and it waits vmcnt(0) before every ds_read while a
        //  global_load_lds is in flight).  The kernel must stay spill-free: a spill of an
        //  in-flight asm destination would save stale data (checked by the build).
        {
            const uint4 *src = w16 + (size_t)(ch * NSLAB) * SLAB_V4;
            for (int i = tid; i < SLAB_V4; i += 512) wbuf[i] = src[i];
        }
        u32x4 wpa[WV4], wpb[WV4];
        auto issue = [&](u32x4 (&wp)[WV4], int slab) {
            const uint4 *src = w16 + (size_t)(ch * NSLAB + slab) * SLAB_V4;
#pragma unroll
            for (int i = 0; i < WV4; ++i) {
                const uint4 *ptr = src + min(tid + i * 512, SLAB_V4 - 1);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wp[i]) : "v"(ptr) : "memory");
            }
        };
        auto park = [&](u32x4 (&wp)[WV4], int slab) {
            u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf + (slab % 3) * SLAB_V4);
#pragma unroll
            for (int i = 0; i < WV4; ++i)
                if (tid + i * 512 < SLAB_V4) dst[tid + i * 512] = wp[i];
        };
        if (NSLAB > 1) issue(wpa, 1);                          // slab 1, parked at the end of step 0
        __syncthreads();
        auto step = [&](int s_, u32x4 (&w_next)[WV4], u32x4 (&w_new)[WV4]) {
            if (s_ + 2 < NSLAB) issue(w_new, s_ + 2);
            const uint4 *wb = wbuf + (s_ % 3) * SLAB_V4 + lane;
            union U { uint4 u; bf16x8 v; };
            U a[TPS][3];
            U b[TPS][3][NT];
            {
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt) {
                    const int tap = s_ * TPS + tt;
                    const int dy = tap / KW, dx = tap - dy * KW;
                    const char *ab;
                    if constexpr (MASKED) {
                        const int rr = lr + dy - PAD_T, cc = lc + dx - PAD_L;
                        const bool inb = (unsigned)rr < (unsigned)p.H && (unsigned)cc < (unsigned)p.W;
                        ab = in_lds + (inb ? abase + ((dy - PAD_T) * RP + (dx - PAD_L)) * BX_PSTRIDE : zero_off);
                    } else {
                        ab = in_lds + abase + (dy * RP + dx) * BX_PSTRIDE;
                    }
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        a[tt][pl].u = *reinterpret_cast<const uint4 *>(ab + pl * 32);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) b[tt][pl][nt].u = wb[((tt * 3 + pl) * NT + nt) * 64];
                    }
                }
            }
            {
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        // smallest terms first
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][2].v, b[tt][0][nt].v, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][1].v, b[tt][1][nt].v, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][0].v, b[tt][2][nt].v, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][1].v, b[tt][0][nt].v, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][0].v, b[tt][1][nt].v, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tt][0].v, b[tt][0][nt].v, acc[nt], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);       // the MFMA chain stays above the slab hand-over
            if (s_ + 1 < NSLAB) {
                // retire slab s+1's loads; slab s+2's (the WV4 newest) may stay in flight
                if (s_ + 2 < NSLAB) {
                    if constexpr (WV4 == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else if constexpr (WV4 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                park(w_next, s_ + 1);
            }
            __syncthreads();
        };
        static_assert(NSLAB % 2 == 0, "slab loop is unrolled by two");
        for (int s_ = 0; s_ < NSLAB; s_ += 2) {
            step(s_, wpa, wpb);
            step(s_ + 1, wpb, wpa);
        }
    }
    // ---- epilogue (as in the f32 kernel, MT = 1) ---------------------------------------------
    const int j = cout_off + (lane & 31);
    float s1[NT], t1[NT], s2[NT], t2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s1[nt] = p.s1[nt * 32 + j]; t1[nt] = p.t1[nt * 32 + j];
        s2[nt] = p.s2 ? p.s2[nt * 32 + j] : 1.f;
        t2[nt] = p.t2 ? p.t2[nt * 32 + j] : 0.f;
    }
    // per batch of EPB rows: all shortcut loads first, then the arithmetic and the stores
    constexpr int EPB = 4;
#pragma unroll
    for (int half = 0; half < 16 / EPB; ++half) {
        int spq[EPB], gwq[EPB];
        float scv[EPB][NT];
#pragma unroll
        for (int e8 = 0; e8 < EPB; ++e8) {
            const int e = half * EPB + e8;
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int q = wid * 32 + row;
            spq[e8] = pos_sp[q];
            gwq[e8] = pos_win[q];
        }
        if (p.sc) {
#pragma unroll
            for (int e8 = 0; e8 < EPB; ++e8) {
                const float *scp = p.sc + (size_t)gwq[e8] * p.sc_win_stride + (size_t)max(spq[e8], 0) * p.cout_total + j;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) scv[e8][nt] = spq[e8] >= 0 ? scp[nt * 32] : 0.f;
            }
        }
#pragma unroll
        for (int e8 = 0; e8 < EPB; ++e8) {
            const int e = half * EPB + e8;
            if (spq[e8] < 0) continue;
            float *o = p.out + (size_t)gwq[e8] * p.out_win_stride + (size_t)spq[e8] * p.cout_total + j;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v = sigmoidf_(acc[nt][e] * s1[nt] + t1[nt]);
                if (p.sc) v = (v + scv[e8][nt]) * s2[nt] + t2[nt];
                o[nt * 32] = v;
            }
        }
    }
}

#include "amt_conv_f16x3.h"

// ---- first layer (Cin = 1): direct convolution on the VALU ---------------------
// A workgroup walks row tiles of 256/COUT*4 output columns of one window row.  The
// input rows (with the zero halo) and all weights sit in LDS; thread = (column
// group, cout) computes 4 consecutive columns x 1 channel, so one weight read
// feeds 4 FMAs and the input reads are wave broadcasts.  Stores are coalesced NHWC.
struct Conv1Params {
    const float *in; size_t in_win_stride;       // [B][H][W]
    float *out; size_t out_win_stride;
    const float *sc; size_t sc_win_stride;
    const float *w;                              // [KH*KW][COUT]
    const float *s1, *t1, *s2, *t2;
    int B, H, W, KH, KW, COUT;
    float *amax_out = nullptr;                   // [B] max |output| per window (atomicMax; conv1_mfma_kernel) or null
};
#define C1_PPT 4
__global__ __launch_bounds__(256) void conv1_kernel(Conv1Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntap = p.KH * p.KW;
    const int groups = 256 / p.COUT;             // column groups per workgroup
    const int TWc = groups * C1_PPT;             // output columns per tile
    const int XW = TWc + p.KW - 1;               // staged input columns
    float *wl = smem;                            // [ntap][COUT]
    float *xt = smem + ntap * p.COUT;            // [KH][XW]
    for (int i = threadIdx.x; i < ntap * p.COUT; i += 256) wl[i] = p.w[i];
    const int pad_t = (p.KH - 1) / 2, pad_l = (p.KW - 1) / 2;
    const int co = threadIdx.x % p.COUT;
    const int pg = threadIdx.x / p.COUT;
    const int tiles_w = (p.W + TWc - 1) / TWc;
    const long total = (long)p.B * p.H * tiles_w;
    const float s1 = p.s1[co], t1 = p.t1[co];
    const float s2 = p.s2 ? p.s2[co] : 1.f, t2 = p.t2 ? p.t2[co] : 0.f;
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int tc = (int)(tile % tiles_w);
        long rest = tile / tiles_w;
        const int r = (int)(rest % p.H);
        const int b = (int)(rest / p.H);
        const int c0 = tc * TWc;
        const float *x = p.in + (size_t)b * p.in_win_stride;
        __syncthreads();                         // previous tile consumed (and wl visible)
        for (int i = threadIdx.x; i < p.KH * XW; i += 256) {
            const int dy = i / XW, cx = i - dy * XW;
            const int gr = r + dy - pad_t, gc = c0 + cx - pad_l;
            xt[i] = (gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? x[(size_t)gr * p.W + gc] : 0.f;
        }
        __syncthreads();
        float acc[C1_PPT];
#pragma unroll
        for (int q = 0; q < C1_PPT; ++q) acc[q] = 0.f;
        for (int dy = 0; dy < p.KH; ++dy) {
            const float *xr = xt + dy * XW + pg * C1_PPT;
            const float *wr = wl + dy * p.KW * p.COUT + co;
            for (int dx = 0; dx < p.KW; ++dx) {
                const float wv = wr[dx * p.COUT];
#pragma unroll
                for (int q = 0; q < C1_PPT; ++q) acc[q] = fmaf(xr[dx + q], wv, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < C1_PPT; ++q) {
            const int c = c0 + pg * C1_PPT + q;
            if (c >= p.W) continue;
            const size_t sp = (size_t)r * p.W + c;
            float v = sigmoidf_(acc[q] * s1 + t1);
            if (p.sc) v = (v + p.sc[(size_t)b * p.sc_win_stride + sp * p.COUT + co]) * s2 + t2;
            p.out[(size_t)b * p.out_win_stride + sp * p.COUT + co] = v;
        }
    }
}

// ---- first layer on the matrix pipe (Cout = 32) ---------------------------------------------
// GEMM view: M = output positions, N = 32 filters, K = KH*KW taps; v_mfma_f32_32x32x2_f32 (f32
// in, f32 accumulate: the same k-ordered fmaf chain as conv1_kernel).  The whole [K][32] kernel
// lives in K/2 B-fragment registers per lane; an A fragment is one ds_read_b32 of the
// single-channel input tile (lane = position, lane half = the odd tap of a tap pair, i.e. the
// next column).  A workgroup (4 waves) owns a TH x TW tile of <= 512 positions = <= 16 M-tiles,
// wave w takes M-tiles w, w+4, ...; workgroups walk the tiles grid-stride.
#define C1M_PCAP 512
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv1_mfma_kernel(Conv1Params p, int TH, int TW, int tiles_h,
                                                          int tiles_w) {
    constexpr int K = KH * KW, NK2 = K / 2;
    static_assert(K % 2 == 0 && KW % 2 == 0, "tap pairs share a kernel row");
    constexpr int PAD_T = (KH - 1) / 2, PAD_L = (KW - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RPW = TW + KW - 1, THin = TH + KH - 1;
    int *pos_rc = reinterpret_cast<int *>(smem);            // [PCAP] r | c << 16 (tile-local)
    int *pos_sp = pos_rc + C1M_PCAP;                         // [PCAP] global spatial index or -1
    float *tpatch = reinterpret_cast<float *>(pos_sp + C1M_PCAP);   // [4 waves][32][HX_TPITCH]
    float *xt = tpatch + 4 * 32 * HX_TPITCH;                 // [THin][RPW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *tb = tpatch + wid * 32 * HX_TPITCH;
    const int c4 = (lane & 7) * 4;
    const int PT = TH * TW, nmt = (PT + 31) >> 5;
    for (int q = tid; q < C1M_PCAP; q += 256) {
        const int r = q / TW, c = q - r * TW;
        pos_rc[q] = q < PT ? (r | (c << 16)) : -1;
    }
    const int co = lane & 31;
    float bq[NK2];
#pragma unroll
    for (int i = 0; i < NK2; ++i) bq[i] = p.w[(2 * i + (lane >> 5)) * 32 + co];
    const float s1 = p.s1[co], t1 = p.t1[co];
    float4 s2v = make_float4(1.f, 1.f, 1.f, 1.f), t2v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.s2) s2v = *reinterpret_cast<const float4 *>(p.s2 + c4);
    if (p.t2) t2v = *reinterpret_cast<const float4 *>(p.t2 + c4);
    const long total = (long)p.B * tiles_h * tiles_w;
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int tc = (int)(tile % tiles_w);
        const long rest = tile / tiles_w;
        const int tr = (int)(rest % tiles_h);
        const int b = (int)(rest / tiles_h);
        const int r0 = tr * TH, c0 = tc * TW;
        const float *x = p.in + (size_t)b * p.in_win_stride;
        __syncthreads();                                     // previous tile consumed (pos_rc visible)
        for (int i = tid; i < THin * RPW; i += 256) {
            const int ri = i / RPW, ci = i - ri * RPW;
            const int gr = r0 + ri - PAD_T, gc = c0 + ci - PAD_L;
            xt[i] = (gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? x[(size_t)gr * p.W + gc] : 0.f;
        }
        for (int q = tid; q < C1M_PCAP; q += 256) {
            const int rc = pos_rc[q];
            const int r = r0 + (rc & 0xFFFF), c = c0 + (rc >> 16);
            pos_sp[q] = (rc >= 0 && r < p.H && c < p.W) ? r * p.W + c : -1;
        }
        __syncthreads();
        float tmax = 0.f;                                    // max |output| of this wave's share of the tile
        for (int mt = wid; mt < nmt; mt += 4) {
            const int rc = pos_rc[mt * 32 + (lane & 31)];
            const int abase = rc >= 0 ? (rc & 0xFFFF) * RPW + (rc >> 16) + (lane >> 5) : (lane >> 5);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int i = 0; i < NK2; ++i) {
                constexpr int dummy = 0; (void)dummy;
                const int dy = (2 * i) / KW, dx = (2 * i) % KW;
                const float a = xt[abase + dy * RPW + dx];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[i], acc, 0, 0, 0);
            }
            // turn the 32 x 32 tile through a wave-private LDS patch: 16-byte stores, 4 channels per lane
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                tb[row * HX_TPITCH + co] = sigmoidf_(acc[e] * s1 + t1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (lane >> 3) + 8 * i;
                const int sp = pos_sp[mt * 32 + row];
                float4 v = *reinterpret_cast<const float4 *>(tb + row * HX_TPITCH + c4);
                if (sp < 0) continue;
                if (p.sc) {
                    const float4 sc = *reinterpret_cast<const float4 *>(p.sc + (size_t)b * p.sc_win_stride + (size_t)sp * 32 + c4);
                    v.x = (v.x + sc.x) * s2v.x + t2v.x; v.y = (v.y + sc.y) * s2v.y + t2v.y;
                    v.z = (v.z + sc.z) * s2v.z + t2v.z; v.w = (v.w + sc.w) * s2v.w + t2v.w;
                }
                tmax = fmaxf(tmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                *reinterpret_cast<float4 *>(p.out + (size_t)b * p.out_win_stride + (size_t)sp * 32 + c4) = v;
            }
        }
        if (p.amax_out) {                                    // one window per tile
            tmax = wave_max(tmax);
            if (lane == 0) atomicMax(reinterpret_cast<int *>(p.amax_out) + b, __float_as_int(tmax));
        }
    }
}
// tile of <= 512 positions that needs the fewest 32-position M-tiles over the image
static void choose_tile1(int H, int W, int *TH_, int *TW_) {
    long best = -1;
    for (int TH = 1; TH <= H && TH <= C1M_PCAP; ++TH) {
        int TWmax = std::min(W, C1M_PCAP / TH);
        for (int TW = std::max(1, TWmax - 40); TW <= TWmax; ++TW) {
            const long tiles = (long)((H + TH - 1) / TH) * ((W + TW - 1) / TW);
            const long cost = tiles * ((TH * TW + 31) / 32);
            if (best < 0 || cost < best) { best = cost; *TH_ = TH; *TW_ = TW; }
        }
    }
}

// ---- shortcut projection: BN(avgpool(conv1x1(x)))  (RDCNN.py:328-334) -----------
// The 1x1 convolution and the average pool commute; pooling first cuts the
// contraction work by the pool area.  A workgroup owns <= 64 output columns of one
// output row: phase 1 pools the inputs into LDS (coalesced over channels), phase 2
// contracts the pooled vectors with the [CIN][COUT] kernel (coalesced over cout).
struct ProjParams {
    const float *in; size_t in_win_stride;       // [B][H][W][CIN]
    float *out; size_t out_win_stride;           // [B][HO][WO][COUT]
    const float *w;                              // [CIN][COUT] or null (identity channels)
    const float *s, *t;                          // folded: out = s*(sum) + t  (bias inside t)
    int B, H, W, CIN, COUT, PH, PW, HO, WO;
};
#define PJ_TW 64
__global__ __launch_bounds__(256) void proj_kernel(ProjParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [PJ_TW][CIN]
    const int wo0 = blockIdx.x * PJ_TW;
    const int ho = blockIdx.y, b = blockIdx.z;
    const int nwo = min(PJ_TW, p.WO - wo0);
    const float inv = 1.0f / (float)(p.PH * p.PW);
    const float *x = p.in + (size_t)b * p.in_win_stride;
    for (int i = threadIdx.x; i < nwo * p.CIN; i += 256) {
        const int wl = i / p.CIN, ci = i - wl * p.CIN;
        const int wo = wo0 + wl;
        float a = 0.f;
        for (int dy = 0; dy < p.PH; ++dy)
            for (int dx = 0; dx < p.PW; ++dx)
                a += x[((size_t)(ho * p.PH + dy) * p.W + (wo * p.PW + dx)) * p.CIN + ci];
        smem[i] = a * inv;
    }
    __syncthreads();
    // four consecutive output channels per thread: 16-byte weight loads and output stores
    float *o = p.out + (size_t)b * p.out_win_stride + ((size_t)ho * p.WO + wo0) * p.COUT;
    const int C4 = p.COUT >> 2;
    for (int i = threadIdx.x; i < nwo * C4; i += 256) {
        const int wl = i / C4, co = (i - wl * C4) << 2;
        float4 acc;
        if (p.w) {
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *pv = smem + wl * p.CIN;
            for (int ci = 0; ci < p.CIN; ++ci) {
                const float4 wv = *reinterpret_cast<const float4 *>(p.w + (size_t)ci * p.COUT + co);
                const float a = pv[ci];
                acc.x = fmaf(a, wv.x, acc.x); acc.y = fmaf(a, wv.y, acc.y);
                acc.z = fmaf(a, wv.z, acc.z); acc.w = fmaf(a, wv.w, acc.w);
            }
        } else {
            acc = *reinterpret_cast<const float4 *>(smem + wl * p.CIN + co);
        }
        const float4 sv = *reinterpret_cast<const float4 *>(p.s + co);
        const float4 tv = *reinterpret_cast<const float4 *>(p.t + co);
        acc.x = acc.x * sv.x + tv.x; acc.y = acc.y * sv.y + tv.y;
        acc.z = acc.z * sv.z + tv.z; acc.w = acc.w * sv.w + tv.w;
        *reinterpret_cast<float4 *>(o + (size_t)wl * p.COUT + co) = acc;
    }
}

// ---- MaxPooling2D (valid, stride = pool) ------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ in,
                                                       size_t in_win_stride, float *__restrict__ out,
                                                       size_t out_win_stride, int B, int H, int W,
                                                       int C, int PH, int PW, int HO, int WO) {
    // four channels per thread (C is a multiple of 32): 16-byte loads and stores
    const int C4 = C >> 2;
    const size_t total = (size_t)B * HO * WO * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) << 2;
        size_t r = i / C4;
        const int wo = (int)(r % WO); r /= WO;
        const int ho = (int)(r % HO);
        const int b = (int)(r / HO);
        const float *x = in + (size_t)b * in_win_stride;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < PH; ++dy)
            for (int dx = 0; dx < PW; ++dx) {
                const float4 v = *reinterpret_cast<const float4 *>(
                    x + ((size_t)(ho * PH + dy) * W + (wo * PW + dx)) * C + c);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        *reinterpret_cast<float4 *>(out + (size_t)b * out_win_stride + ((size_t)ho * WO + wo) * C + c) = m;
    }
}

// ---- Dense: C[M][N] = act(A[M][K] B[K][N] + bias[N]) on v_mfma_f32_32x32x2_f32 -----
// M = windows is small (<= 512 per chunk) and N = 300, so the grid of output tiles is
// tiny; to fill the chip a workgroup owns one 32x32 output tile and its 4 waves split K
// four ways (wave-private LDS staging, [32][33] pitch: conflict-free fragment reads), then
// the four partial tiles are summed through LDS in a fixed order (deterministic).  A long
// contraction (the 5120-wide flatten of the timing head) is additionally split over
// gridDim.z workgroups that write partial tiles; dense_reduce_kernel adds them in z order.
#define DN_KC 32
#define DN_KSPLIT 8
__global__ __launch_bounds__(256) void dense_kernel(const float *__restrict__ A, int K,
                                                     const float *__restrict__ Bm,
                                                     const float *__restrict__ bias, int N,
                                                     float *__restrict__ Cm, int M, int act,
                                                     float *__restrict__ part) {
    __shared__ float as[4][32 * 33];
    __shared__ float bs[4][DN_KC * 32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int KS = gridDim.z;
    const int Kz = ((K + KS - 1) / KS + 4 * DN_KC - 1) / (4 * DN_KC) * (4 * DN_KC);   // K range per workgroup
    const int kz1 = min(K, (int)(blockIdx.z + 1) * Kz);
    const int kq = Kz / 4;                                           // K range per wave, chunk aligned
    const int kbeg = blockIdx.z * Kz + wid * kq, kend = min(kz1, kbeg + kq);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float *aw = as[wid], *bw = bs[wid];
    for (int k0 = 0; k0 < kq; k0 += DN_KC) {                         // same trip count in every wave
        __syncthreads();
        for (int i = lane; i < 32 * DN_KC; i += 64) {
            const int r = i / DN_KC, kk = i - r * DN_KC;
            const int k = kbeg + k0 + kk;
            aw[r * 33 + kk] = (m0 + r < M && k < kend) ? A[(size_t)(m0 + r) * K + k] : 0.f;
        }
        for (int i = lane; i < DN_KC * 32; i += 64) {
            const int kk = i >> 5, c = i & 31;
            const int k = kbeg + k0 + kk;
            bw[i] = (k < kend && n0 + c < N) ? Bm[(size_t)k * N + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < DN_KC; kk += 2) {
            const float a = aw[(lane & 31) * 33 + kk + (lane >> 5)];
            const float bv = bw[(kk + (lane >> 5)) * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    // partial tiles -> LDS [wave][row][col(33)], then every thread sums 4 outputs
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        as[wid][row * 33 + (lane & 31)] = acc[e];
    }
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 256) {
        const int row = i >> 5, col = i & 31;
        const int m = m0 + row, n = n0 + col;
        if (m < M && n < N) {
            float v = ((as[0][row * 33 + col] + as[1][row * 33 + col]) +
                       (as[2][row * 33 + col] + as[3][row * 33 + col]));
            if (part) { part[((size_t)blockIdx.z * M + m) * N + n] = v; continue; }
            v += bias[n];
            if (act == 1) v = sigmoidf_(v);
            Cm[(size_t)m * N + n] = v;
        }
    }
}
__global__ void dense_reduce_kernel(const float *__restrict__ part, int KS, const float *__restrict__ bias,
                                    int N, float *__restrict__ Cm, int M, int act) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    float v = part[i];
    for (int z = 1; z < KS; ++z) v += part[(size_t)z * M * N + i];
    v += bias[i % N];
    if (act == 1) v = sigmoidf_(v);
    Cm[i] = v;
}

// ---- output activation: softmax (K > 1) or sigmoid + range scaling (K == 1) -------
__global__ void head_output_kernel(const float *__restrict__ logits, float *__restrict__ y, int B,
                                   int K, float lo, float hi) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *l = logits + (size_t)b * K;
    float *o = y + (size_t)b * K;
    if (K == 1) {
        const float a = 1.0f / (1.0f + expf(-l[0]));
        o[0] = a * (hi - lo) + lo;                 // RDCNN.py:308-310 with out_func range [0,1]
    } else {
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, l[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(l[k] - m);
        for (int k = 0; k < K; ++k) o[k] = expf(l[k] - m) / s;
    }
}

// =====================================================================================
// Host side: topology walk, weight folding / pre-arrangement, launch plan
// =====================================================================================
struct DevBuf { float *p = nullptr; };

struct ConvOp {
    int cin, cout, H, W, kh, kw;
    float *w = nullptr, *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;
    bool residual = false;
    int sc_proj = -1;          // index into projs, -1 => identity shortcut
    int pool_after = 0;        // 1 => maxpool (ph, pw) follows
    int TH = 0, TW = 0, NWIN = 1, MT = 2;
    size_t lds = 0;
    int nslice = 1, cw = 0;    // output channels are computed in nslice slices of cw channels
    // split-bf16 variant (null when not built for this layer)
    uint4 *w16 = nullptr;
    int TH16 = 0, TW16 = 0, NWIN16 = 1;
    size_t lds16 = 0;
    int nslice16 = 1, cw16 = 0;
    bool masked16 = false;
    double eff32 = 0, eff16 = 0;
    // split-fp16 variant (null when not built for this layer); N-slicing as for split-bf16
    uint4 *whs = nullptr;      // 32-wide N-slices: 16x16x32 fragments of tap pairs (conv_f16x3s_kernel)
    int nsliceh = 1, cwh = 0;  // split-fp16 N-slicing: 32-wide slices unless the layer runs the masked form
    int THh = 0, TWh = 0, NWINh = 1;
    size_t ldsh = 0;
    bool maskedh = false;
    double effh = 0;
    int sw = 0;                // weights scaled by 2^sw
    // FFT-domain form (conv mode 3, amt_fftconv.hip): built on the first amt_rdcnn_set_mode(net, 3) for the
    // 32 -> 32 (4 x 16) layers on images of at most 561 columns; k_host keeps the Keras-layout kernel for that
    std::vector<float> k_host;
    amt_fftconv_layer *fft = nullptr;
    // packed-image form of the 64 -> 64 (4 x 16) layers on 10 x 64 images (amt_fftpk.hip), same switch
    amt_fftpk_layer *pk = nullptr;
};
struct ProjOp {
    int cin, cout, H, W, ph, pw, HO, WO;
    float *w = nullptr, *s = nullptr, *t = nullptr;
};
struct Tower {
    int in_h, in_w, ph, pw;
    std::vector<ConvOp> convs;
    std::vector<ProjOp> projs;
    size_t max_act = 0;        // floats per window of the largest activation
    int out_h, out_w, out_c;
};
struct ProfPending { hipEvent_t e0, e1; int tower, layer, windows; };
struct ProfAcc { double ms = 0, windows = 0; long launches = 0; };
struct amt_rdcnn {
    // optional per-conv-launch timing (HIP events on the caller's stream)
    mutable bool prof_on = false;
    mutable std::vector<ProfPending> prof_pending;
    mutable std::vector<hipEvent_t> prof_free;
    mutable std::vector<std::vector<ProfAcc>> prof_acc;   // [tower][layer]
    amt_rdcnn_desc d;
    std::vector<Tower> towers;
    std::vector<float *> allocs;
    float *d1w = nullptr, *d1b = nullptr, *d2w = nullptr, *d2b = nullptr;
    int flat = 0;
    double flops = 0;
    mutable int mode = 0;      // 0: f32 MFMA, 1: split-bf16 where built, 2: split-fp16 where built
};

static int upload(amt_rdcnn *n, const std::vector<float> &h, float **out) {
    float *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return AMT_E_NOMEM;
    n->allocs.push_back(d);
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return AMT_E_HIP;
    *out = d;
    return AMT_OK;
}

struct BN { const float *g, *b, *m, *v; };
static void fold_bn(const BN &bn, int c, const float *bias, std::vector<float> &s, std::vector<float> &t) {
    s.resize(c); t.resize(c);
    for (int i = 0; i < c; ++i) {
        const float sc = bn.g[i] / sqrtf(bn.v[i] + RD_BN_EPS);
        s[i] = sc;
        t[i] = bn.b[i] + ((bias ? bias[i] : 0.f) - bn.m[i]) * sc;
    }
}

static void choose_tile(ConvOp &c) {
    double best = -1;
    for (int MT = 2; MT >= 1; --MT) {
        const int pcap = 128 * MT;
        auto consider = [&](int TH, int TW, int NWIN) {
            const size_t posin = (size_t)NWIN * (TH + c.kh - 1) * (TW + c.kw - 1);
            const size_t lds = posin * RD_CSTRIDE * 4 + 2 * (size_t)RD_CC * c.cw * 4 + (size_t)pcap * 8;
            if (lds > 150 * 1024) return;
            const double tiles = (double)((c.H + TH - 1) / TH) * ((c.W + TW - 1) / TW) / NWIN;
            double eff = (double)c.H * c.W / (tiles * pcap);
            if (lds > 78 * 1024) eff *= 0.93;          // prefer two workgroups per CU
            if (eff > best + 1e-9) { best = eff; c.TH = TH; c.TW = TW; c.NWIN = NWIN; c.MT = MT; c.lds = lds; c.eff32 = eff; }
        };
        if (c.H * c.W <= pcap) {
            for (int nw = pcap / (c.H * c.W); nw >= 1; --nw) consider(c.H, c.W, nw);
        }
        for (int TH = 1; TH <= c.H && TH <= pcap; ++TH) {
            int TW = pcap / TH;
            if (TW > c.W) TW = c.W;
            if (TW >= 1) consider(TH, TW, 1);
            // also the narrowest TW that keeps the same number of column tiles
            const int nct = (c.W + TW - 1) / TW;
            const int TW2 = (c.W + nct - 1) / nct;
            if (TW2 >= 1 && TW2 <= TW) consider(TH, TW2, 1);
        }
    }
}


static size_t slab16_bytes(const ConvOp &c) {
    const int NT = c.cw16 / 32, ntaps = c.kh * c.kw;
    const int tps = (NT == 1 && ntaps % 2 == 0) ? 2 : 1;
    return (size_t)tps * 3 * NT * 64 * 16;
}
static void choose_tile16(ConvOp &c) {
    double best = -1;
    const int pcap = 256;
    c.masked16 = false;
    if (c.H * c.W <= 64) {
        // small image: halo-free (masked) tile of whole images
        const int nw = pcap / (c.H * c.W);
        const size_t lds = ((size_t)nw * c.H * c.W + 1) * BX_PSTRIDE + 3 * slab16_bytes(c) + (size_t)pcap * 8;
        if (lds <= 79 * 1024) {
            c.masked16 = true; c.TH16 = c.H; c.TW16 = c.W; c.NWIN16 = nw; c.lds16 = lds;
            c.eff16 = (double)nw * c.H * c.W / pcap;
            return;
        }
    }
    auto consider = [&](int TH, int TW, int NWIN) {
        const size_t posin = (size_t)NWIN * (TH + c.kh - 1) * bx_row_pitch(TW, TW + c.kw - 1);
        const size_t lds = posin * BX_PSTRIDE + 3 * slab16_bytes(c) + (size_t)pcap * 8;
        if (lds > 79 * 1024) return;                    // two workgroups per CU
        const double tiles = (double)((c.H + TH - 1) / TH) * ((c.W + TW - 1) / TW) / NWIN;
        const double eff = (double)c.H * c.W / (tiles * pcap);
        if (eff > best + 1e-9) { best = eff; c.TH16 = TH; c.TW16 = TW; c.NWIN16 = NWIN; c.lds16 = lds; c.eff16 = eff; }
    };
    if (c.H * c.W <= pcap)
        for (int nw = pcap / (c.H * c.W); nw >= 1; --nw) consider(c.H, c.W, nw);
    for (int TH = 1; TH <= c.H && TH <= pcap; ++TH) {
        int TW = pcap / TH;
        if (TW > c.W) TW = c.W;
        if (TW >= 1) consider(TH, TW, 1);
        const int nct = (c.W + TW - 1) / TW;
        const int TW2 = (c.W + nct - 1) / nct;
        if (TW2 >= 1 && TW2 <= TW) consider(TH, TW2, 1);
    }
}

template <int KH, int KW, int CIN, int COUT, bool MASKED>
static int launch_conv16_t(const ConvOp &c, ConvParams p, hipStream_t st) {
    auto kern = conv_bf16x6_kernel<KH, KW, CIN, COUT, MASKED>;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)));
        attr_set = true;
    }
    p.TH = c.TH16; p.TW = c.TW16; p.NWIN = c.NWIN16;
    p.tiles_h = (p.H + p.TH - 1) / p.TH; p.tiles_w = (p.W + p.TW - 1) / p.TW;
    const int groups = (p.B + p.NWIN - 1) / p.NWIN;
    const unsigned grid = (unsigned)((size_t)groups * p.tiles_h * p.tiles_w);
    kern<<<dim3(grid, c.nslice16), 512, c.lds16, st>>>(p, c.w16);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
template <int KH, int KW>
static int launch_conv16_k(const ConvOp &c, const ConvParams &p, hipStream_t st) {
#define BX_CASE(CI, CO)                                                              \
    if (c.cin == CI && c.cw16 == CO)                                                 \
        return c.masked16 ? launch_conv16_t<KH, KW, CI, CO, true>(c, p, st)          \
                          : launch_conv16_t<KH, KW, CI, CO, false>(c, p, st);
    BX_CASE(32, 32) BX_CASE(32, 64) BX_CASE(64, 64) BX_CASE(128, 64)
#undef BX_CASE
    return AMT_E_UNSUPPORTED;
}
static bool conv16_supported(const ConvOp &c) {
    return (c.cin == 32 && c.cout == 32) || (c.cin == 32 && c.cout == 64) || (c.cin == 64 && c.cout == 64) ||
           (c.cin == 64 && c.cout == 128) || (c.cin == 128 && c.cout == 128);
}
static int launch_conv16(const ConvOp &c, const ConvParams &p, hipStream_t st) {
    if (c.kh == 4 && c.kw == 16) return launch_conv16_k<4, 16>(c, p, st);
    if (c.kh == 4 && c.kw == 2) return launch_conv16_k<4, 2>(c, p, st);
    if (c.kh == 2 && c.kw == 2) return launch_conv16_k<2, 2>(c, p, st);
    return AMT_E_UNSUPPORTED;
}


// ---- split-fp16 variant: tile choice and launch ---------------------------------------------
#define HX_SLAB_BYTES 4096
#define HX_XCHG_BYTES (8 * 32 * HX_TPITCH * 4)  // epilogue transposition patches (>= the K-split exchange, 8 x 4 KB)
static void choose_tile_h(ConvOp &c) {
    double best = -1;
    const int pcap = 256;
    c.maskedh = false; c.THh = 0;
    // two weight-group buffers (a group = min(4, steps per chunk) steps of 4 KB)
    const int NTh = c.cwh / 32;
    const int nsteps = c.kh * c.kw / (NTh == 1 ? 2 : 1);
    const size_t wbytes = 2 * (size_t)std::min(4, nsteps) * HX_SLAB_BYTES;
    auto total = [&](size_t posbytes) {
        return std::max(posbytes, (size_t)HX_XCHG_BYTES) + wbytes + (size_t)pcap * 12 + (size_t)HX_MAXWIN * 8 + 512;      // + the slice's folded BN parameters
    };
    if (c.H * c.W <= 64) {
        const int nw = std::min(pcap / (c.H * c.W), HX_MAXWIN);
        const size_t lds = total(((size_t)nw * c.H * c.W + 1) * HX_PSTRIDE);
        if (lds <= 79 * 1024) {
            c.maskedh = true; c.THh = c.H; c.TWh = c.W; c.NWINh = nw; c.ldsh = lds;
            c.effh = (double)nw * c.H * c.W / pcap;
            return;
        }
    }
    auto consider = [&](int TH, int TW, int NWIN) {
        const size_t posin = (size_t)NWIN * (TH + c.kh - 1) * bx_row_pitch(TW, TW + c.kw - 1);
        const size_t lds = total(posin * HX_PSTRIDE);
        if (lds > 79 * 1024) return;                    // two workgroups per CU
        const double tiles = (double)((c.H + TH - 1) / TH) * ((c.W + TW - 1) / TW) / NWIN;
        const double eff = (double)c.H * c.W / (tiles * pcap);
        if (eff > best + 1e-9) { best = eff; c.THh = TH; c.TWh = TW; c.NWINh = NWIN; c.ldsh = lds; c.effh = eff; }
    };
    if (c.H * c.W <= pcap)
        for (int nw = std::min(pcap / (c.H * c.W), HX_MAXWIN); nw >= 1; --nw) consider(c.H, c.W, nw);
    for (int TH = 1; TH <= c.H && TH <= pcap; ++TH) {
        int TW = pcap / TH;
        if (TW > c.W) TW = c.W;
        if (TW >= 1) consider(TH, TW, 1);
        const int nct = (c.W + TW - 1) / TW;
        const int TW2 = (c.W + nct - 1) / nct;
        if (TW2 >= 1 && TW2 <= TW) consider(TH, TW2, 1);
    }
}

// AMT_CONV_MS=4 (diagnostic, 4 x 16 kernels only): four M-subtiles per wave -- 64 positions, 256 threads, 12 LDS
// fragment reads per 24 MFMAs instead of 8 per 12.  Parity-green and 3 % SLOWER than the default (385 vs 397 TFLOP/s
// on the 20 x 516 layers): the kernel is not bound by LDS reads, and two waves per SIMD hide less.
static int conv_ms_choice() {
    static int ms = -1;
    if (ms < 0) {
        const char *e = getenv("AMT_CONV_MS");
        ms = (e && atoi(e) == 4) ? 4 : 2;
    }
    return ms;
}
template <int KH, int KW, int CIN, bool MASKED, int MS = 2>
static int launch_convs_t(const ConvOp &c, ConvParams p, const float *amax_in, float *amax_out, hipStream_t st) {
    auto kern = conv_f16x3s_kernel<KH, KW, CIN, MASKED, MS>;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)));
        attr_set = true;
    }
    p.TH = c.THh; p.TW = c.TWh; p.NWIN = c.NWINh;
    p.tiles_h = (p.H + p.TH - 1) / p.TH; p.tiles_w = (p.W + p.TW - 1) / p.TW;
    const int groups = (p.B + p.NWIN - 1) / p.NWIN;
    const unsigned grid = (unsigned)((size_t)groups * p.tiles_h * p.tiles_w);
    HxScale hs{amax_in, amax_out, c.sw, nullptr};
    static const bool want_ts = getenv("AMT_CONV_TS") != nullptr;      // diagnostic: phase split of a workgroup's life
    static unsigned long long *ts_dev = nullptr;
    static size_t ts_cap = 0;
    const size_t nwg = (size_t)grid * c.nsliceh;
    if (want_ts) {
        if (nwg > ts_cap) {
            if (ts_dev) (void)hipFree(ts_dev);
            AMT_HIP_CHECK(hipMalloc(&ts_dev, nwg * 4 * sizeof(unsigned long long)));
            ts_cap = nwg;
        }
        hs.ts = ts_dev;
    }
    kern<<<dim3(grid, c.nsliceh), 1024 / MS, c.ldsh, st>>>(p, c.whs, hs);
    AMT_LAUNCH_CHECK();
    if (want_ts) {
        AMT_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(nwg * 4);
        AMT_HIP_CHECK(hipMemcpy(h.data(), ts_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double a = 0, b = 0, e = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < nwg; ++i) {
            a += (double)(h[4 * i + 1] - h[4 * i]); b += (double)(h[4 * i + 2] - h[4 * i + 1]);
            e += (double)(h[4 * i + 3] - h[4 * i + 2]);
            t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 3]);
        }
        fprintf(stderr, "conv_ts k%dx%d cin%d %dx%d masked%d wgs %zu: start->first MFMA %.2f us, MFMA loop %.2f us, epilogue %.2f us; "
                        "kernel %.1f us = %.2f workgroup lives per slot of 512\n",
                KH, KW, CIN, p.H, p.W, (int)MASKED, nwg, a / nwg / 100.0, b / nwg / 100.0, e / nwg / 100.0,
                (double)(t1 - t0) / 100.0, (double)(t1 - t0) * 512.0 / ((a + b + e)));
    }
    return AMT_OK;
}
template <int KH, int KW>
static int launch_convh_k(const ConvOp &c, const ConvParams &p, const float *amax_in, float *amax_out, hipStream_t st) {
    // every layer runs 32-wide N-slices (blockIdx.y) of the single-tile kernel; small images its masked form
    if (!c.whs || c.cwh != 32) return AMT_E_UNSUPPORTED;
#define HXS_CASE(CI)                                                                       \
    if (c.cin == CI) {                                                                     \
        if (c.maskedh) return launch_convs_t<KH, KW, CI, true>(c, p, amax_in, amax_out, st);              \
        if constexpr (KW == 16)                                                            \
            if (conv_ms_choice() == 4) return launch_convs_t<KH, KW, CI, false, 4>(c, p, amax_in, amax_out, st); \
        return launch_convs_t<KH, KW, CI, false>(c, p, amax_in, amax_out, st);            \
    }
    HXS_CASE(32) HXS_CASE(64) HXS_CASE(128)
#undef HXS_CASE
    return AMT_E_UNSUPPORTED;
}
static int launch_convh(const ConvOp &c, const ConvParams &p, const float *amax_in, float *amax_out, hipStream_t st) {
    if (c.kh == 4 && c.kw == 16) return launch_convh_k<4, 16>(c, p, amax_in, amax_out, st);
    if (c.kh == 4 && c.kw == 2) return launch_convh_k<4, 2>(c, p, amax_in, amax_out, st);
    if (c.kh == 2 && c.kw == 2) return launch_convh_k<2, 2>(c, p, amax_in, amax_out, st);
    return AMT_E_UNSUPPORTED;
}

template <int KH, int KW, int CIN, int COUT, int MT>
static int launch_conv_t(const ConvOp &c, const ConvParams &p, hipStream_t st) {
    auto kern = conv_mfma_kernel<KH, KW, CIN, COUT, MT>;
    static size_t attr_set = 0;
    if (c.lds > attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(152 * 1024)));
        attr_set = 152 * 1024;
    }
    const int groups = (p.B + p.NWIN - 1) / p.NWIN;
    const unsigned grid = (unsigned)((size_t)groups * p.tiles_h * p.tiles_w);
    kern<<<dim3(grid, c.nslice), 256, c.lds, st>>>(p);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

template <int KH, int KW>
static int launch_conv_k(const ConvOp &c, const ConvParams &p, hipStream_t st) {
#define RD_CASE(CI, CO)                                                         \
    if (c.cin == CI && c.cw == CO)                                              \
        return c.MT == 2 ? launch_conv_t<KH, KW, CI, CO, 2>(c, p, st)           \
                         : launch_conv_t<KH, KW, CI, CO, 1>(c, p, st);
    RD_CASE(32, 32) RD_CASE(32, 64) RD_CASE(64, 64) RD_CASE(64, 128) RD_CASE(128, 128)
    RD_CASE(64, 32) RD_CASE(128, 32)
#undef RD_CASE
    return AMT_E_UNSUPPORTED;
}

static int launch_conv(const ConvOp &c, const ConvParams &p, hipStream_t st) {
    if (c.kh == 4 && c.kw == 16) return launch_conv_k<4, 16>(c, p, st);
    if (c.kh == 4 && c.kw == 2) return launch_conv_k<4, 2>(c, p, st);
    if (c.kh == 2 && c.kw == 2) return launch_conv_k<2, 2>(c, p, st);
    return AMT_E_UNSUPPORTED;
}

static bool conv_supported(int kh, int kw) {
    return (kh == 4 && kw == 16) || (kh == 4 && kw == 2) || (kh == 2 && kw == 2);
}

// number of f32 the canonical blob must hold; also validates the topology
static long walk_count(const amt_rdcnn_desc &d, int *flat_out, std::string *err) {
    long n = 0;
    int flat = 0;
    for (int t = 0; t < d.n_towers; ++t) {
        int H = d.in_h[t], W = d.in_w[t], C = 1, fo = 32;
        int p0H = H, p0W = W, p0C = 1;
        for (int i = 1; i <= d.conv_layers; ++i) {
            n += (long)d.kh[t] * d.kw[t] * C * fo + fo + 4 * fo;
            C = fo;
            if (d.residual_frequency > 0 && i % d.residual_frequency == 0) {
                if (!(p0H == H && p0W == W && p0C == C)) {
                    if (p0C != C) n += (long)p0C * C + C;
                    n += 4 * C;
                }
                n += 4 * C;
                p0H = H; p0W = W; p0C = C;
            }
            if (d.pool_layer_frequency > 0 && i % d.pool_layer_frequency == 0) {
                H /= d.pool_h[t]; W /= d.pool_w[t];
                if (H < 1 || W < 1) { if (err) *err = "pooling collapses the activation"; return -1; }
            }
            if (d.feature_expand_frequency > 0 && i % d.feature_expand_frequency == 0) fo *= 2;
        }
        flat += H * W * C;
    }
    n += (long)flat * d.dense_units + d.dense_units;
    n += (long)d.dense_units * d.output_classes + d.output_classes;
    if (flat_out) *flat_out = flat;
    return n;
}

extern "C" {

size_t amt_rdcnn_param_count(const amt_rdcnn_desc *desc) {
    if (!desc || desc->n_towers < 1 || desc->n_towers > 2) return 0;
    long n = walk_count(*desc, nullptr, nullptr);
    return n < 0 ? 0 : (size_t)n;
}

int amt_rdcnn_destroy(amt_rdcnn *net) {
    if (!net) return AMT_OK;
    for (const ProfPending &pp : net->prof_pending) { (void)hipEventDestroy(pp.e0); (void)hipEventDestroy(pp.e1); }
    for (hipEvent_t e : net->prof_free) (void)hipEventDestroy(e);
    for (float *p : net->allocs) (void)hipFree(p);
    for (Tower &t : net->towers)
        for (ConvOp &c : t.convs)
        {
            if (c.fft) amt_fftconv_layer_destroy_internal(c.fft);
            if (c.pk) amt_fftpk_layer_destroy_internal(c.pk);
        }
    delete net;
    return AMT_OK;
}

int amt_rdcnn_create(amt_rdcnn **out, const amt_rdcnn_desc *desc, const float *wh, size_t n_floats) {
    if (!out || !desc || !wh) return AMT_E_INVALID;
    const amt_rdcnn_desc &d = *desc;
    if (d.n_towers < 1 || d.n_towers > 2 || d.conv_layers < 1 || d.dense_units < 1 ||
        d.output_classes < 1)
        return AMT_E_INVALID;
    int flat = 0;
    const long need = walk_count(d, &flat, nullptr);
    if (need < 0) return AMT_E_UNSUPPORTED;
    if ((size_t)need != n_floats) return AMT_E_SHAPE;
    amt_rdcnn *n = new amt_rdcnn();
    n->d = d;
    n->flat = flat;
    const float *cur = wh;
    auto take = [&](size_t k) { const float *r = cur; cur += k; return r; };
    int rc = AMT_OK;
#define RD_TRY(x) do { rc = (x); if (rc != AMT_OK) { amt_rdcnn_destroy(n); return rc; } } while (0)
    for (int t = 0; t < d.n_towers; ++t) {
        Tower tw;
        tw.in_h = d.in_h[t]; tw.in_w = d.in_w[t]; tw.ph = d.pool_h[t]; tw.pw = d.pool_w[t];
        int H = tw.in_h, W = tw.in_w, C = 1, fo = 32;
        int p0H = H, p0W = W, p0C = 1;
        tw.max_act = (size_t)H * W;
        const int kh = d.kh[t], kw = d.kw[t];
        for (int i = 1; i <= d.conv_layers; ++i) {
            ConvOp c;
            c.cin = C; c.cout = fo; c.H = H; c.W = W; c.kh = kh; c.kw = kw;
            const float *kern = take((size_t)kh * kw * C * fo);
            const float *bias = take(fo);
            BN bn{take(fo), take(fo), take(fo), take(fo)};
            std::vector<float> s, tt;
            fold_bn(bn, fo, bias, s, tt);
            RD_TRY(upload(n, s, &c.s1));
            RD_TRY(upload(n, tt, &c.t1));
            if (C == 1) {
                std::vector<float> w1(kern, kern + (size_t)kh * kw * fo);    // [tap][cout] as is
                RD_TRY(upload(n, w1, &c.w));
            } else {
                if (C % 32 || fo % 32 || !conv_supported(kh, kw) ||
                    !((C == 32 && fo == 32) || (C == 32 && fo == 64) || (C == 64 && fo == 64) ||
                      (C == 64 && fo == 128) || (C == 128 && fo == 128))) {
                    amt_rdcnn_destroy(n);
                    return AMT_E_UNSUPPORTED;
                }
                if ((kh == 4 && kw == 16 && C == 32 && fo == 32 && W + 15 <= 576 && H <= 20) ||
                    (kh == 4 && kw == 16 && C == 64 && fo == 64 && W == 64 && H == 10))
                    c.k_host.assign(kern, kern + (size_t)kh * kw * C * fo);      // FFT-domain forms, built on demand (mode 3)
                const int NT = fo / 32, nch = C / 32, ntap = kh * kw;
                // small late-stage layers (few output positions per window) cannot fill 256 CUs with
                // position tiles alone: compute them in 32-channel output slices (blockIdx.y)
                c.nslice = (fo >= 128 && H * W <= 128) ? fo / 32 : 1;
                c.cw = fo / c.nslice;
                const int NTW = c.cw / 32;
                std::vector<float> wa((size_t)ntap * C * fo);
                // [slice][cchunk][tap][c][j][ntw]  <-  keras [tap][cin][cout], cout = slice*cw + 32*ntw + j
                for (int sl = 0; sl < c.nslice; ++sl)
                    for (int ch = 0; ch < nch; ++ch)
                        for (int tap = 0; tap < ntap; ++tap)
                            for (int cc = 0; cc < 32; ++cc)
                                for (int j = 0; j < 32; ++j)
                                    for (int nt = 0; nt < NTW; ++nt)
                                        wa[(((((size_t)sl * nch + ch) * ntap + tap) * 32 + cc) * 32 + j) * NTW + nt] =
                                            kern[((size_t)tap * C + ch * 32 + cc) * fo + sl * c.cw + nt * 32 + j];
                RD_TRY(upload(n, wa, &c.w));
                choose_tile(c);
                if (c.TH == 0) { amt_rdcnn_destroy(n); return AMT_E_UNSUPPORTED; }
                if (conv16_supported(c)) {
                    c.nslice16 = fo > 64 ? fo / 64 : 1;                // 128 couts run as two 64-wide slices
                    c.cw16 = fo / c.nslice16;
                    choose_tile16(c);
                    // worth it only if 2.67x fewer matrix cycles survive the tile efficiency
                    if (c.TH16 > 0 && c.eff16 * 2.67 > c.eff32 * 1.15) {
                        const int NT16 = c.cw16 / 32;
                        const int tps = (NT16 == 1 && ntap % 2 == 0) ? 2 : 1;
                        const int nslab = ntap / tps;
                        // [slice][chunk16][slab][tt][plane][nt][h][col][8] bf16
                        const int nch16 = C / BX_CC;
                        std::vector<unsigned short> w16((size_t)nch16 * ntap * 3 * NT * 2 * 32 * 8);
                        for (int sl = 0; sl < c.nslice16; ++sl)
                            for (int ch = 0; ch < nch16; ++ch)
                                for (int sb = 0; sb < nslab; ++sb)
                                    for (int tt = 0; tt < tps; ++tt)
                                        for (int nt = 0; nt < NT16; ++nt)
                                            for (int h = 0; h < 2; ++h)
                                                for (int col = 0; col < 32; ++col)
                                                    for (int jj = 0; jj < 8; ++jj) {
                                                        const int tap = sb * tps + tt;
                                                        const int cin_i = ch * BX_CC + 8 * h + jj;
                                                        const float wv = kern[((size_t)tap * C + cin_i) * fo + sl * c.cw16 + nt * 32 + col];
                                                        unsigned short hh[3];
                                                        amt_split3(wv, hh[0], hh[1], hh[2]);
                                                        for (int pl = 0; pl < 3; ++pl) {
                                                            const size_t idx =
                                                                ((((((((size_t)sl * nch16 + ch) * nslab + sb) * tps + tt) * 3 + pl) * NT16 + nt) * 2 + h) * 32 + col) * 8 + jj;
                                                            w16[idx] = hh[pl];
                                                        }
                                                    }
                        void *d16 = nullptr;
                        if (hipMalloc(&d16, w16.size() * 2) != hipSuccess) { amt_rdcnn_destroy(n); return AMT_E_NOMEM; }
                        n->allocs.push_back(static_cast<float *>(d16));
                        if (hipMemcpy(d16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                            amt_rdcnn_destroy(n); return AMT_E_HIP;
                        }
                        c.w16 = static_cast<uint4 *>(d16);
                    }
                    // split-fp16 weights, scaled 2^sw (layout below)
                    c.cwh = 32; c.nsliceh = fo / 32;                   // 32-wide N-slices
                    choose_tile_h(c);
                    float wmax = 0.f;
                    bool finite = true;
                    for (size_t q = 0; q < (size_t)ntap * C * fo; ++q) {
                        if (!std::isfinite(kern[q])) finite = false;
                        wmax = std::max(wmax, fabsf(kern[q]));
                    }
                    if (c.THh > 0 && finite && wmax > 0.f) {
                        int ew = 0;
                        (void)frexpf(wmax, &ew);                          // wmax < 2^ew
                        c.sw = 4 - ew;                                     // max |w| 2^sw in [8, 16)
                        const float wscale = ldexpf(1.0f, c.sw);
                        const int nch16 = C / BX_CC;
                        if (c.cwh == 32) {
                            // [slice][chunk16][tap pair][plane][N-subtile][lane = col + 16 kgroup][8] f16:
                            // kgroup g = (tap 2 tp + g % 2, channels 8 (g / 2) .. + 7)
                            std::vector<unsigned short> ws((size_t)nch16 * ntap * 2 * NT * 2 * 32 * 8);
                            const int ntp = ntap / 2;
                            for (int sl = 0; sl < c.nsliceh; ++sl)
                                for (int ch = 0; ch < nch16; ++ch)
                                    for (int tp = 0; tp < ntp; ++tp)
                                        for (int ns = 0; ns < 2; ++ns)
                                            for (int ln = 0; ln < 64; ++ln)
                                                for (int jj = 0; jj < 8; ++jj) {
                                                    const int col = ln & 15, kg = ln >> 4;
                                                    const int tap = 2 * tp + (kg & 1);
                                                    const int cin_i = ch * BX_CC + 8 * (kg >> 1) + jj;
                                                    const float wv = kern[((size_t)tap * C + cin_i) * fo + sl * 32 + ns * 16 + col];
                                                    unsigned short hh[2];
                                                    amt_split_f16<true>(wv * wscale, hh[0], hh[1]);
                                                    for (int pl = 0; pl < 2; ++pl) {
                                                        const size_t idx =
                                                            ((((((size_t)sl * nch16 + ch) * ntp + tp) * 2 + pl) * 2 + ns) * 64 + ln) * 8 + jj;
                                                        ws[idx] = hh[pl];
                                                    }
                                                }
                            void *ds = nullptr;
                            if (hipMalloc(&ds, ws.size() * 2) != hipSuccess) { amt_rdcnn_destroy(n); return AMT_E_NOMEM; }
                            n->allocs.push_back(static_cast<float *>(ds));
                            if (hipMemcpy(ds, ws.data(), ws.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                                amt_rdcnn_destroy(n); return AMT_E_HIP;
                            }
                            c.whs = static_cast<uint4 *>(ds);
                        }
                    }
                }
            }
            n->flops += 2.0 * H * W * (double)kh * kw * C * fo;
            C = fo;
            tw.max_act = std::max(tw.max_act, (size_t)H * W * C);
            if (d.residual_frequency > 0 && i % d.residual_frequency == 0) {
                c.residual = true;
                if (!(p0H == H && p0W == W && p0C == C)) {
                    ProjOp pr;
                    pr.cin = p0C; pr.cout = C; pr.H = p0H; pr.W = p0W;
                    pr.ph = p0H / H; pr.pw = p0W / W;            // floor(sh1/sh2), RDCNN.py:325-326
                    pr.HO = p0H / pr.ph; pr.WO = p0W / pr.pw;    // valid avg-pool output
                    if (pr.HO != H || pr.WO != W) { amt_rdcnn_destroy(n); return AMT_E_UNSUPPORTED; }
                    const float *pk = nullptr, *pb = nullptr;
                    if (p0C != C) { pk = take((size_t)p0C * C); pb = take(C); }
                    BN sbn{take(C), take(C), take(C), take(C)};
                    std::vector<float> ps, pt;
                    fold_bn(sbn, C, pb, ps, pt);
                    if (pk) { std::vector<float> pw_(pk, pk + (size_t)p0C * C); RD_TRY(upload(n, pw_, &pr.w)); }
                    RD_TRY(upload(n, ps, &pr.s));
                    RD_TRY(upload(n, pt, &pr.t));
                    n->flops += 2.0 * H * W * (double)p0C * C;
                    c.sc_proj = (int)tw.projs.size();
                    tw.projs.push_back(pr);
                }
                BN rbn{take(C), take(C), take(C), take(C)};
                std::vector<float> rs, rt;
                fold_bn(rbn, C, nullptr, rs, rt);
                RD_TRY(upload(n, rs, &c.s2));
                RD_TRY(upload(n, rt, &c.t2));
                p0H = H; p0W = W; p0C = C;
            }
            if (d.pool_layer_frequency > 0 && i % d.pool_layer_frequency == 0) {
                c.pool_after = 1;
                H /= tw.ph; W /= tw.pw;
            }
            if (d.feature_expand_frequency > 0 && i % d.feature_expand_frequency == 0) fo *= 2;
            tw.convs.push_back(c);
        }
        tw.out_h = H; tw.out_w = W; tw.out_c = C;
        n->towers.push_back(tw);
    }
    {
        const float *k1 = take((size_t)flat * d.dense_units);
        const float *b1 = take(d.dense_units);
        const float *k2 = take((size_t)d.dense_units * d.output_classes);
        const float *b2 = take(d.output_classes);
        std::vector<float> a(k1, b1), b(b1, k2), c2(k2, b2), e(b2, cur);
        RD_TRY(upload(n, a, &n->d1w));
        RD_TRY(upload(n, b, &n->d1b));
        RD_TRY(upload(n, c2, &n->d2w));
        RD_TRY(upload(n, e, &n->d2b));
        n->flops += 2.0 * flat * d.dense_units + 2.0 * d.dense_units * d.output_classes;
    }
#undef RD_TRY
    if ((size_t)(cur - wh) != n_floats) { amt_rdcnn_destroy(n); return AMT_E_SHAPE; }
    *out = n;
    return AMT_OK;
}

double amt_rdcnn_flops_per_window(const amt_rdcnn *net) { return net ? net->flops : 0.0; }

int amt_rdcnn_set_mode(amt_rdcnn *net, int mode) {
    if (!net || mode < 0 || mode > 3) return AMT_E_INVALID;
    if (mode == 3) {
        // FFT-domain form of the 32 -> 32 (4 x 16) layers (every other layer runs the split-fp16 kernels of mode 2):
        // the transformed kernel matrices are computed on the host in float64, once
        for (Tower &t : net->towers)
            for (ConvOp &c : t.convs)
                if (!c.fft && !c.pk && !c.k_host.empty()) {
                    const int rc = c.cin == 64 ? amt_fftpk_layer_create_internal(&c.pk, c.k_host.data())
                                               : amt_fftconv_layer_create_internal(&c.fft, c.k_host.data());
                    if (rc != AMT_OK) return rc;
                }
    }
    net->mode = mode;
    return AMT_OK;
}

int amt_rdcnn_profile(amt_rdcnn *net, int enable) {
    if (!net) return AMT_E_INVALID;
    net->prof_on = enable != 0;
    if (net->prof_acc.empty()) {
        net->prof_acc.resize(net->towers.size());
        for (size_t t = 0; t < net->towers.size(); ++t) net->prof_acc[t].resize(net->towers[t].convs.size());
    }
    return AMT_OK;
}

int amt_rdcnn_profile_read(amt_rdcnn *net, int32_t *desc, double *ms, double *windows,
                           double *flops_per_window, int cap, int *n_rows, int reset) {
    if (!net || !n_rows) return AMT_E_INVALID;
    for (const ProfPending &pp : net->prof_pending) {
        AMT_HIP_CHECK(hipEventSynchronize(pp.e1));
        float e = 0.f;
        AMT_HIP_CHECK(hipEventElapsedTime(&e, pp.e0, pp.e1));
        ProfAcc &a = net->prof_acc[pp.tower][pp.layer];
        a.ms += e; a.windows += pp.windows; a.launches += 1;
        net->prof_free.push_back(pp.e0);
        net->prof_free.push_back(pp.e1);
    }
    net->prof_pending.clear();
    int r = 0;
    for (size_t t = 0; t < net->prof_acc.size(); ++t)
        for (size_t i = 0; i < net->prof_acc[t].size(); ++i) {
            if (r < cap && desc && ms && windows && flops_per_window) {
                const ConvOp &c = net->towers[t].convs[i];
                int32_t *d = desc + (size_t)r * 8;
                d[0] = (int)t; d[1] = (int)i + 1; d[2] = c.kh; d[3] = c.kw; d[4] = c.cin; d[5] = c.cout;
                d[6] = c.H; d[7] = c.W;
                ms[r] = net->prof_acc[t][i].ms;
                windows[r] = net->prof_acc[t][i].windows;
                flops_per_window[r] = 2.0 * c.H * c.W * (double)c.kh * c.kw * c.cin * c.cout;
            }
            ++r;
        }
    *n_rows = r;
    if (reset)
        for (auto &v : net->prof_acc) for (auto &a : v) a = ProfAcc();
    return AMT_OK;
}

#define RD_CHUNK_MAX 1024
// windows per pass through the network (AMT_RD_CHUNK, diagnostic: smaller chunks keep the FFT-domain layers' frequency
// tensors inside the Infinity Cache at the price of more, smaller launches)
static int rd_chunk() {
    static int c = 0;
    if (!c) { const char *e = getenv("AMT_RD_CHUNK"); c = e ? std::max(1, std::min(atoi(e), RD_CHUNK_MAX)) : RD_CHUNK_MAX; }
    return c;
}
#define RD_CHUNK rd_chunk()
static size_t ws_floats(const amt_rdcnn *n, int Bc) {
    size_t ma = 0;
    for (const Tower &t : n->towers) ma = std::max(ma, t.max_act);
    ma = (ma + 3) & ~(size_t)3;
    // + per-window max |activation| of the network input and of every conv layer's output (split-fp16 scaling)
    size_t fft = 0;                                      // mode 3: two frequency tensors + per-window max |Xf|
    if (n->mode == 3)
        for (const Tower &t : n->towers)
            for (const ConvOp &c : t.convs)
            {
                if (c.fft) fft = std::max(fft, 2 * amt_fftconv_freq_floats(Bc, c.H) + (size_t)Bc + 8);
                if (c.pk) fft = std::max(fft, 2 * amt_fftpk_freq_floats(Bc) + (size_t)Bc + 8);
            }
    return (size_t)Bc * (4 * ma + (size_t)((n->flat + 3) & ~3) + (size_t)((n->d.dense_units + 3) & ~3) +
                         (size_t)((n->d.output_classes + 3) & ~3) + (size_t)(n->d.conv_layers + 1)) + 8 + fft;
}

size_t amt_rdcnn_workspace_bytes(const amt_rdcnn *net, int B) {
    if (!net || B <= 0) return 0;
    return ws_floats(net, std::min(B, RD_CHUNK)) * sizeof(float);
}

static int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)std::min<size_t>(g, 65536);
}

int amt_rdcnn_forward(const amt_rdcnn *net, const float *const *x, int B, float *y, float *logits,
                      void *workspace, size_t workspace_bytes, void *stream) {
    if (!net || !x || !y || !workspace || B <= 0) return AMT_E_INVALID;
    const amt_rdcnn_desc &d = net->d;
    for (int t = 0; t < d.n_towers; ++t) if (!x[t]) return AMT_E_INVALID;
    if (workspace_bytes < amt_rdcnn_workspace_bytes(net, B)) return AMT_E_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    size_t ma = 0;
    for (const Tower &t : net->towers) ma = std::max(ma, t.max_act);
    ma = (ma + 3) & ~(size_t)3;
    const int K = d.output_classes, DU = d.dense_units, flat = net->flat;

    for (int b0 = 0; b0 < B; b0 += RD_CHUNK) {
        const int Bc = std::min(RD_CHUNK, B - b0);
        float *ws = static_cast<float *>(workspace);
        float *buf[4];
        for (int i = 0; i < 4; ++i) buf[i] = ws + (size_t)i * Bc * ma;
        float *flatbuf = ws + (size_t)4 * Bc * ma;
        float *d1 = flatbuf + (size_t)Bc * ((flat + 3) & ~3);
        float *lg = d1 + (size_t)Bc * ((DU + 3) & ~3);
        float *amax = lg + (size_t)Bc * ((K + 3) & ~3);              // [conv_layers + 1][Bc] max |activation| per window
        int flat_off = 0;
        for (int t = 0; t < d.n_towers; ++t) {
            const Tower &tw = net->towers[t];
            const float *cur = x[t] + (size_t)b0 * tw.in_h * tw.in_w;
            if (net->mode >= 2) {
                AMT_HIP_CHECK(hipMemsetAsync(amax, 0, (size_t)(d.conv_layers + 1) * Bc * sizeof(float), st));
                const size_t nin = (size_t)tw.in_h * tw.in_w;
                absmax_kernel<<<dim3((unsigned)std::min<size_t>((nin + 1023) / 1024, 64), Bc), 256, 0, st>>>(cur, nin, nin, amax);
            }
            size_t cur_stride = (size_t)tw.in_h * tw.in_w;
            const float *p0 = cur; size_t p0_stride = cur_stride;
            int H = tw.in_h, W = tw.in_w;
            const int L = (int)tw.convs.size();
            bool xf_valid = false;                       // mode 3: the previous layer left its output transformed in Xf
            auto pick = [&](const float *a, const float *b_, const float *c_) -> float * {
                for (int i = 0; i < 4; ++i)
                    if (buf[i] != a && buf[i] != b_ && buf[i] != c_) return buf[i];
                return nullptr;
            };
            for (int i = 0; i < L; ++i) {
                const ConvOp &c = tw.convs[i];
                const bool last_op = (i == L - 1) && !c.pool_after;
                float *o = last_op ? flatbuf + flat_off : pick(cur, p0, nullptr);
                const size_t o_stride = last_op ? (size_t)flat : (size_t)H * W * c.cout;
                const float *sc = nullptr; size_t sc_stride = 0;
                const ProjOp *rank1 = nullptr;
                if (c.residual) {
                    if (c.sc_proj >= 0 && ((net->mode == 2 && c.whs && !c.maskedh) || (net->mode == 3 && (c.fft || (c.whs && !c.maskedh)))) &&
                        tw.projs[c.sc_proj].cin == 1 &&
                        tw.projs[c.sc_proj].ph == 1 && tw.projs[c.sc_proj].pw == 1 && tw.projs[c.sc_proj].w) {
                        rank1 = &tw.projs[c.sc_proj];                 // formed in the consumer's epilogue
                    } else if (c.sc_proj >= 0) {
                        const ProjOp &pr = tw.projs[c.sc_proj];
                        float *sb = pick(cur, p0, o);
                        ProjParams pp{p0, p0_stride, sb, (size_t)pr.HO * pr.WO * pr.cout, pr.w, pr.s, pr.t,
                                      Bc, pr.H, pr.W, pr.cin, pr.cout, pr.ph, pr.pw, pr.HO, pr.WO};
                        proj_kernel<<<dim3((pr.WO + PJ_TW - 1) / PJ_TW, pr.HO, Bc), 256,
                                      (size_t)PJ_TW * pr.cin * sizeof(float), st>>>(pp);
                        sc = sb; sc_stride = (size_t)pr.HO * pr.WO * pr.cout;
                    } else {
                        sc = p0; sc_stride = p0_stride;
                    }
                }
                hipEvent_t pe0 = nullptr, pe1 = nullptr;
                if (net->prof_on) {
                    for (hipEvent_t *pe : {&pe0, &pe1}) {
                        if (!net->prof_free.empty()) { *pe = net->prof_free.back(); net->prof_free.pop_back(); }
                        else AMT_HIP_CHECK(hipEventCreate(pe));
                    }
                    AMT_HIP_CHECK(hipEventRecord(pe0, st));
                }
                // split-fp16 scaling: layer i reads amax[i] (its input) and leaves amax[i + 1] (its output)
                float *amax_o = net->mode >= 2 ? amax + (size_t)(i + 1) * Bc : nullptr;
                bool wrote_amax = false;
                if (c.cin == 1 && c.cout == 32 && conv_supported(c.kh, c.kw)) {
                    Conv1Params cp{cur, cur_stride, o, o_stride, sc, sc_stride, c.w, c.s1, c.t1,
                                   c.residual ? c.s2 : nullptr, c.residual ? c.t2 : nullptr,
                                   Bc, H, W, c.kh, c.kw, c.cout, amax_o};
                    wrote_amax = true;
                    int TH1 = 1, TW1 = 1;
                    choose_tile1(H, W, &TH1, &TW1);
                    const int th = (H + TH1 - 1) / TH1, twn = (W + TW1 - 1) / TW1;
                    const size_t lds = (size_t)2 * C1M_PCAP * 4 + (size_t)4 * 32 * HX_TPITCH * 4 +
                                       (size_t)(TH1 + c.kh - 1) * (TW1 + c.kw - 1) * 4;
                    const unsigned grid = (unsigned)std::min<size_t>((size_t)Bc * th * twn, 256 * 8);
                    if (c.kh == 4 && c.kw == 16) conv1_mfma_kernel<4, 16><<<grid, 256, lds, st>>>(cp, TH1, TW1, th, twn);
                    else if (c.kh == 4 && c.kw == 2) conv1_mfma_kernel<4, 2><<<grid, 256, lds, st>>>(cp, TH1, TW1, th, twn);
                    else conv1_mfma_kernel<2, 2><<<grid, 256, lds, st>>>(cp, TH1, TW1, th, twn);
                    AMT_LAUNCH_CHECK();
                } else if (c.cin == 1) {
                    Conv1Params cp{cur, cur_stride, o, o_stride, sc, sc_stride, c.w, c.s1, c.t1,
                                   c.residual ? c.s2 : nullptr, c.residual ? c.t2 : nullptr,
                                   Bc, H, W, c.kh, c.kw, c.cout};
                    const int groups = 256 / c.cout;
                    const size_t lds = ((size_t)c.kh * c.kw * c.cout +
                                        (size_t)c.kh * (groups * C1_PPT + c.kw - 1)) * 4;
                    const size_t tiles = (size_t)Bc * H * ((W + groups * C1_PPT - 1) / (groups * C1_PPT));
                    conv1_kernel<<<(unsigned)std::min<size_t>(tiles, 8192), 256, lds, st>>>(cp);
                    AMT_LAUNCH_CHECK();
                } else if (net->mode == 3 && c.fft) {
                    // FFT-domain form: [forward transform of the spatial input, unless the previous layer left its
                    // output transformed] -> per-frequency GEMM -> inverse + epilogue [+ forward transform for the next
                    // such layer]; the spatial output is written only where something reads it (a later shortcut, a
                    // pooling, a layer of another kind)
                    float *Xf = amax + (size_t)(d.conv_layers + 1) * Bc + 8;
                    Xf = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(Xf) + 15) & ~(uintptr_t)15);
                    float *Yf = Xf + amt_fftconv_freq_floats(Bc, H);
                    float *amaxf = Yf + amt_fftconv_freq_floats(Bc, H);
                    int rc = AMT_OK;
                    if (!xf_valid) rc = amt_fftconv_forward_fft(c.fft, cur, cur_stride, Bc, H, W, Xf, amaxf, st);
                    if (rc == AMT_OK) rc = amt_fftconv_gemm(c.fft, Xf, amaxf, Bc, H, Yf, st);
                    if (rc != AMT_OK) return rc;
                    const bool next_fft = i + 1 < L && !c.pool_after && tw.convs[i + 1].fft;
                    const bool need_sp = !next_fft || c.residual;
                    FcEpilogue ep;
                    ep.s1 = c.s1; ep.t1 = c.t1;
                    if (c.residual) {
                        ep.s2 = c.s2; ep.t2 = c.t2;
                        if (rank1) { ep.sc1 = p0; ep.sc1_stride = p0_stride; ep.sc1_w = rank1->w; ep.sc1_s = rank1->s; ep.sc1_t = rank1->t; }
                        else { ep.sc = sc; ep.sc_stride = sc_stride; }
                    }
                    rc = amt_fftconv_inverse_epilogue(c.fft, Yf, ep, Bc, H, W, need_sp ? o : nullptr, o_stride,
                                                      next_fft ? Xf : nullptr, amaxf, next_fft ? nullptr : amax_o, st);
                    if (rc != AMT_OK) return rc;
                    xf_valid = next_fft;
                    wrote_amax = true;
                } else if (net->mode == 3 && c.pk) {
                    // packed-image form of a 10 x 64, 64 -> 64 layer: the same chaining, one GEMM row per window
                    float *Xf = amax + (size_t)(d.conv_layers + 1) * Bc + 8;
                    Xf = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(Xf) + 15) & ~(uintptr_t)15);
                    float *Yf = Xf + amt_fftpk_freq_floats(Bc);
                    float *amaxf = Yf + amt_fftpk_freq_floats(Bc);
                    int rc = AMT_OK;
                    if (!xf_valid) rc = amt_fftpk_forward_fft(c.pk, cur, cur_stride, Bc, Xf, amaxf, st);
                    if (rc == AMT_OK) rc = amt_fftpk_gemm(c.pk, Xf, amaxf, Bc, Yf, st);
                    if (rc != AMT_OK) return rc;
                    const bool next_pk = i + 1 < L && !c.pool_after && tw.convs[i + 1].pk;
                    const bool need_sp = !next_pk || c.residual;
                    FcEpilogue ep;
                    ep.s1 = c.s1; ep.t1 = c.t1;
                    if (c.residual) { ep.s2 = c.s2; ep.t2 = c.t2; ep.sc = sc; ep.sc_stride = sc_stride; }
                    rc = amt_fftpk_inverse_epilogue(c.pk, Yf, ep, Bc, need_sp ? o : nullptr, o_stride, next_pk ? Xf : nullptr, amaxf,
                                                    next_pk ? nullptr : amax_o, st);
                    if (rc != AMT_OK) return rc;
                    xf_valid = next_pk;
                    wrote_amax = true;
                } else {
                    ConvParams cp{cur, cur_stride, o, o_stride, sc, sc_stride, c.w, c.s1, c.t1,
                                  c.residual ? c.s2 : nullptr, c.residual ? c.t2 : nullptr,
                                  Bc, H, W, c.TH, c.TW, c.NWIN, (H + c.TH - 1) / c.TH,
                                  (W + c.TW - 1) / c.TW, c.cout};
                    if (rank1) {
                        cp.sc1 = p0; cp.sc1_win_stride = p0_stride;
                        cp.sc1_w = rank1->w; cp.sc1_s = rank1->s; cp.sc1_t = rank1->t;
                    }
                    const bool fp16 = net->mode >= 2 && c.whs;
                    const int rc = fp16 ? launch_convh(c, cp, amax + (size_t)i * Bc, amax_o, st)
                                   : (net->mode >= 1 && c.w16) ? launch_conv16(c, cp, st) : launch_conv(c, cp, st);
                    if (rc != AMT_OK) return rc;
                    wrote_amax = fp16;
                }
                if (amax_o && !wrote_amax && i + 1 < L) {
                    // the producing kernel does not measure its output: one extra pass (layers the split-fp16
                    // kernel is not built for; none of the reference's head topologies)
                    const size_t nout = (size_t)H * W * c.cout;
                    absmax_kernel<<<dim3((unsigned)std::min<size_t>((nout + 1023) / 1024, 64), Bc), 256, 0, st>>>(
                        o, nout, o_stride, amax_o);
                }
                if (net->prof_on) {
                    AMT_HIP_CHECK(hipEventRecord(pe1, st));
                    net->prof_pending.push_back(ProfPending{pe0, pe1, t, i, Bc});
                }
                if (c.residual) { p0 = o; p0_stride = o_stride; }
                cur = o; cur_stride = o_stride;
                if (c.pool_after) {
                    const int HO = H / tw.ph, WO = W / tw.pw;
                    const bool last = (i == L - 1);
                    float *po = last ? flatbuf + flat_off : pick(cur, p0, nullptr);
                    const size_t po_stride = last ? (size_t)flat : (size_t)HO * WO * c.cout;
                    maxpool_kernel<<<grid_for((size_t)Bc * HO * WO * c.cout / 4), 256, 0, st>>>(
                        cur, cur_stride, po, po_stride, Bc, H, W, c.cout, tw.ph, tw.pw, HO, WO);
                    cur = po; cur_stride = po_stride; H = HO; W = WO;
                }
            }
            flat_off += tw.out_h * tw.out_w * tw.out_c;
        }
        if (flat >= 2048 && ma >= (size_t)DN_KSPLIT * DU) {
            float *dpart = buf[0];                         // the activation buffers are dead by now
            dense_kernel<<<dim3((DU + 31) / 32, (Bc + 31) / 32, DN_KSPLIT), 256, 0, st>>>(
                flatbuf, flat, net->d1w, net->d1b, DU, d1, Bc, 1, dpart);
            dense_reduce_kernel<<<(unsigned)(((size_t)Bc * DU + 255) / 256), 256, 0, st>>>(
                dpart, DN_KSPLIT, net->d1b, DU, d1, Bc, 1);
        } else {
            dense_kernel<<<dim3((DU + 31) / 32, (Bc + 31) / 32), 256, 0, st>>>(
                flatbuf, flat, net->d1w, net->d1b, DU, d1, Bc, 1, nullptr);
        }
        dense_kernel<<<dim3((K + 31) / 32, (Bc + 31) / 32), 256, 0, st>>>(
            d1, DU, net->d2w, net->d2b, K, lg, Bc, 0, nullptr);
        head_output_kernel<<<(Bc + 63) / 64, 64, 0, st>>>(lg, y + (size_t)b0 * K, Bc, K, d.out_lo, d.out_hi);
        if (logits)
            AMT_HIP_CHECK(hipMemcpyAsync(logits + (size_t)b0 * K, lg, (size_t)Bc * K * sizeof(float),
                                         hipMemcpyDeviceToDevice, st));
        AMT_LAUNCH_CHECK();
    }
    return AMT_OK;
}

}  // extern "C"

// ---- trainer form of the split-fp16 kernels (amt_convh.h) --------------------------------------------------------
int amt_convh_plan_init(amt_convh_plan *pl, int kh, int kw, int cin, int cout, int H, int W) {
    if (!pl || !conv_supported(kh, kw)) return AMT_E_UNSUPPORTED;
    if (!(cin == 32 || cin == 64 || cin == 128) || cout < 32 || cout % 32 != 0 || H < 1 || W < 1) return AMT_E_UNSUPPORTED;
    ConvOp c;
    c.cin = cin; c.cout = cout; c.H = H; c.W = W; c.kh = kh; c.kw = kw;
    c.cwh = 32; c.nsliceh = cout / 32;
    choose_tile_h(c);
    if (c.THh <= 0) return AMT_E_UNSUPPORTED;
    pl->kh = kh; pl->kw = kw; pl->cin = cin; pl->cout = cout; pl->H = H; pl->W = W;
    pl->TH = c.THh; pl->TW = c.TWh; pl->NWIN = c.NWINh; pl->masked = c.maskedh ? 1 : 0; pl->lds = c.ldsh;
    return AMT_OK;
}
size_t amt_convh_packed_bytes(const amt_convh_plan *pl) {
    return (size_t)pl->kh * pl->kw * pl->cin * pl->cout * 2 /* planes */ * 2 /* bytes */;
}

__global__ __launch_bounds__(256) void convh_wmax_kernel(const amt_convh_pack_job *jobs) {
    __shared__ float red[16];
    const amt_convh_pack_job j = jobs[blockIdx.y];
    const size_t n = (size_t)j.ntap * j.Cin * j.Cout;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(j.w[i]));
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int *>(j.wmax), __float_as_int(m));
}
// one thread per (slice, chunk, tap pair, N-subtile, lane): eight channels x two planes = two 16-byte stores.
// Layout (the host loop of amt_rdcnn_create, "split-fp16 weights"): [slice][chunk16][tap pair][plane][N-subtile][lane][8]
__global__ __launch_bounds__(256) void convh_pack_kernel(const amt_convh_pack_job *jobs) {
    const amt_convh_pack_job j = jobs[blockIdx.y];
    const float wmax = *j.wmax;
    int sw = 0;
    if (wmax > 0.f && wmax < 3.0e38f) {
        int ew = 0;
        (void)frexpf(wmax, &ew);
        sw = 4 - ew;
    }
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) *j.sw = sw;
    const float wscale = ldexpf(1.0f, sw);
    const bool bwd = blockIdx.z == 1;
    uint4 *out = static_cast<uint4 *>(bwd ? j.packed_bwd : j.packed_fwd);
    if (!out) return;
    const int C = bwd ? j.Cout : j.Cin, fo = bwd ? j.Cin : j.Cout;          // channels of the convolution that runs
    const int nch16 = C / BX_CC, ntp = j.ntap / 2, nsl = fo / 32;
    const size_t total = (size_t)nsl * nch16 * ntp * 2 * 64;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int ln = (int)(q & 63);
        size_t r = q >> 6;
        const int ns = (int)(r & 1); r >>= 1;
        const int tp = (int)(r % ntp); r /= ntp;
        const int ch = (int)(r % nch16);
        const int sl = (int)(r / nch16);
        const int col = ln & 15, kg = ln >> 4;
        const int tap = 2 * tp + (kg & 1);
        const int co = sl * 32 + ns * 16 + col;
        unsigned short h[2][8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int ci = ch * BX_CC + 8 * (kg >> 1) + jj;
            const float wv = bwd ? j.w[((size_t)(j.ntap - 1 - tap) * j.Cin + co) * j.Cout + ci]
                                 : j.w[((size_t)tap * j.Cin + ci) * j.Cout + co];
            amt_split_f16<true>(wv * wscale, h[0][jj], h[1][jj]);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint4 pk;
            pk.x = h[pl][0] | ((unsigned)h[pl][1] << 16);
            pk.y = h[pl][2] | ((unsigned)h[pl][3] << 16);
            pk.z = h[pl][4] | ((unsigned)h[pl][5] << 16);
            pk.w = h[pl][6] | ((unsigned)h[pl][7] << 16);
            out[((((((size_t)sl * nch16 + ch) * ntp + tp) * 2 + pl) * 2 + ns) * 64) + ln] = pk;
        }
    }
}
int amt_convh_pack_all(const amt_convh_pack_job *jobs_dev, int n, int max_elems, hipStream_t st) {
    if (n <= 0) return AMT_OK;
    const unsigned gx = (unsigned)std::min<size_t>(((size_t)max_elems + 256 * 16 - 1) / (256 * 16), 64);
    convh_wmax_kernel<<<dim3(gx, n), 256, 0, st>>>(jobs_dev);
    convh_pack_kernel<<<dim3(gx, n, 2), 256, 0, st>>>(jobs_dev);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
int amt_convh_absmax(const float *x, size_t n, int B, float *amax, hipStream_t st) {
    absmax_kernel<<<dim3((unsigned)std::min<size_t>((n + 1023) / 1024, 64), B), 256, 0, st>>>(x, n, n, amax);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
template <int KH, int KW, int CIN, bool MASKED>
static int launch_convtr_t(const amt_convh_plan &pl, const ConvParams &p, const void *packed, const HxScale &hs, hipStream_t st) {
    auto kern = conv_f16x3s_kernel<KH, KW, CIN, MASKED, 2, true>;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(80 * 1024)));
        attr_set = true;
    }
    const int groups = (p.B + p.NWIN - 1) / p.NWIN;
    const unsigned grid = (unsigned)((size_t)groups * p.tiles_h * p.tiles_w);
    kern<<<dim3(grid, pl.cout / 32), 512, pl.lds, st>>>(p, static_cast<const uint4 *>(packed), hs);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
template <int KH, int KW>
static int launch_convtr_k(const amt_convh_plan &pl, const ConvParams &p, const void *packed, const HxScale &hs, hipStream_t st) {
#define HXT_CASE(CI)                                                                            \
    if (pl.cin == CI)                                                                           \
        return pl.masked ? launch_convtr_t<KH, KW, CI, true>(pl, p, packed, hs, st)             \
                         : launch_convtr_t<KH, KW, CI, false>(pl, p, packed, hs, st);
    HXT_CASE(32) HXT_CASE(64) HXT_CASE(128)
#undef HXT_CASE
    return AMT_E_UNSUPPORTED;
}
int amt_convh_run(const amt_convh_plan *pl, const float *in, float *out, const float *acc, int B, const void *packed,
                  const int *sw_dev, const float *bias, const float *amax_in, int pad_t, int pad_l, hipStream_t st) {
    if (!pl || !in || !out || !packed || !sw_dev || !amax_in || B <= 0) return AMT_E_INVALID;
    if (pad_t < 0 || pad_t >= pl->kh || pad_l < 0 || pad_l >= pl->kw) return AMT_E_INVALID;
    const size_t istr = (size_t)pl->H * pl->W * pl->cin, ostr = (size_t)pl->H * pl->W * pl->cout;
    ConvParams p{in, istr, out, ostr, acc, ostr, nullptr, nullptr, bias, nullptr, nullptr,
                 B, pl->H, pl->W, pl->TH, pl->TW, pl->NWIN, (pl->H + pl->TH - 1) / pl->TH, (pl->W + pl->TW - 1) / pl->TW, pl->cout};
    p.pad_t = pad_t; p.pad_l = pad_l;
    HxScale hs{amax_in, nullptr, 0, nullptr};
    hs.sw_dev = sw_dev;
    if (pl->kh == 4 && pl->kw == 16) return launch_convtr_k<4, 16>(*pl, p, packed, hs, st);
    if (pl->kh == 4 && pl->kw == 2) return launch_convtr_k<4, 2>(*pl, p, packed, hs, st);
    if (pl->kh == 2 && pl->kw == 2) return launch_convtr_k<2, 2>(*pl, p, packed, hs, st);
    return AMT_E_UNSUPPORTED;
}
