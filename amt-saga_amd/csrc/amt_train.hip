// RDCNN training step for gfx950: forward in training mode, backward, Adagrad.
//
// Replaces, for the graph of /root/reference/RDCNN.py:176-233, what ``res_net.train`` /
// ``res_net.test`` run through Keras (RDCNN.py:503-526, :559-589): ``model.train_on_batch`` /
// ``test_on_batch`` of a model compiled with ``keras.optimizers.Adagrad()`` and
// ``mean_squared_error`` (one output) or ``sparse_categorical_crossentropy`` (RDCNN.py:245-254).
// CPU restatement and its pinning (PyTorch autograd): oracle/train.py, tests/test_oracle_train.py.
//
// Training runs at the reference's batch sizes (8 windows, main.py -batch_size), far from the
// inference loop's thousands, so this file favours one small set of exact, deterministic kernels
// over per-layer fusion:
//   * every contraction -- convolution forward (im2col rows x kernel), its data gradient
//     (dZ x kernel^T -> col2im) and its weight gradient (im2col^T x dZ), the Dense layers and their
//     gradients -- is ONE tiled GEMM kernel on the f32 matrix pipe (v_mfma_f32_32x32x2_f32, 64 x 64 tile
//     per workgroup, operands staged through LDS, optional transposed operands, split over the
//     contraction with a fixed-order reduction: no float atomics anywhere);
//   * BatchNormalization in training mode: per-channel batch statistics by two-stage column reductions,
//     moving statistics updated with momentum 0.99; its backward needs two more column sums;
//   * sigmoid, shortcut add, max / average pooling and their gradients are elementwise / gather kernels;
//   * Adagrad (a += g^2, p -= lr g / (sqrt(a) + eps); accumulators start at 0 as in Keras 2.2 / tf.keras 1.13 or
//     at 0.1 as in tf.keras >= 1.14 -- the reference pins neither) is one elementwise kernel per tensor.
// A small tape (list of ops over numbered tensors) is built from the topology descriptor once; a step
// walks it forwards, then backwards accumulating gradients per tensor (the shortcut source receives two).
#include "amt_common.h"
#include "amt_convh.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

typedef float tf32x16 __attribute__((ext_vector_type(16)));
#define TR_BN_EPS 1e-3f
#define TR_BN_MOMENTUM 0.99f
#define TG_TM 64
#define TG_TN 64
#define TG_KC 32

// ---- GEMM: C[M][N] = sum_k A(m,k) B(k,n) (+ bias[n]); A stored [M][lda] (or [K][lda] when TA),
//      B stored [K][ldb] (or [N][ldb] when TB).  gridDim.z slices of the contraction write partial
//      tiles to `part` ([z][M][N]); splitk_reduce_kernel adds them in z order.
// WN = 2: 64 x 64 tile, waves 2 x 2;  WN = 1: 128 x 32 tile, waves 4 x 1 (the convolutions' N is 32 channels half
// of the time: a 64-wide tile would leave every second MFMA column block empty)
template <bool TA, bool TB, int WN>
__global__ __launch_bounds__(256) void tgemm_kernel(const float *__restrict__ A, int lda,
                                                     const float *__restrict__ B, int ldb,
                                                     float *__restrict__ C, int ldc, int M, int N, int K,
                                                     int kslice, const float *__restrict__ bias,
                                                     float *__restrict__ part) {
    constexpr int TM = 32 * (4 / WN), TN = 32 * WN;
    __shared__ float As[TM][TG_KC + 1];
    __shared__ float Bs[TG_KC][TN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int kbeg = blockIdx.z * kslice, kend = min(K, kbeg + kslice);
    const int mi = WN == 2 ? wid >> 1 : wid, ni = WN == 2 ? wid & 1 : 0;
    tf32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int NA = (TM * TG_KC) / 256, NB = (TG_KC * TN) / 256;
    float ra[NA], rb[NB];
    // the next k-step's tiles are requested before this step's MFMAs and parked in LDS after them: the global
    // latency runs under the matrix work instead of in front of it
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256;
            int r, c;
            if (TA) { c = idx / TM; r = idx - c * TM; } else { r = idx / TG_KC; c = idx - r * TG_KC; }
            const int m = m0 + r, k = k0 + c;
            ra[i] = (m < M && k < kend) ? (TA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            int c, j;
            if (TB) { j = idx / TG_KC; c = idx - j * TG_KC; } else { c = idx / TN; j = idx - c * TN; }
            const int k = k0 + c, n = n0 + j;
            rb[i] = (n < N && k < kend) ? (TB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n]) : 0.f;
        }
    };
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TG_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256;
            int r, c;
            if (TA) { c = idx / TM; r = idx - c * TM; } else { r = idx / TG_KC; c = idx - r * TG_KC; }
            As[r][c] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            int c, j;
            if (TB) { j = idx / TG_KC; c = idx - j * TG_KC; } else { c = idx / TN; j = idx - c * TN; }
            Bs[c][j] = rb[i];
        }
        __syncthreads();
        if (k0 + TG_KC < kend) gload(k0 + TG_KC);
#pragma unroll
        for (int kk = 0; kk < TG_KC; kk += 2) {
            const float a = As[mi * 32 + (lane & 31)][kk + (lane >> 5)];
            const float b = Bs[kk + (lane >> 5)][ni * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int n = n0 + ni * 32 + (lane & 31);
    if (n >= N) return;
    const float bv = (bias && !part) ? bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        if (part) part[((size_t)blockIdx.z * M + m) * N + n] = acc[e];
        else C[(size_t)m * ldc + n] = acc[e] + bv;
    }
}
__global__ void splitk_reduce_kernel(const float *__restrict__ part, int Z, float *__restrict__ C, int ldc,
                                     int M, int N, const float *__restrict__ bias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t MN = (size_t)M * N;
    if (i >= MN) return;
    float v = part[i];
    int z = 1;
    for (; z + 8 <= Z; z += 8) {                             // eight loads in flight, added in z order
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = part[(size_t)(z + u) * MN + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += q[u];
    }
    for (; z < Z; ++z) v += part[(size_t)z * MN + i];
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    C[(size_t)m * ldc + n] = v + (bias ? bias[n] : 0.f);
}

// ---- im2col / col2im (Keras "same" padding: before = (k-1)/2, the rest after) -------------------
// One workgroup per output position (b, h, w): its K = KH KW C row is written contiguously, four channels per
// lane and access where C allows (the index arithmetic is 32-bit and per tap, not 64-bit per element).
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x, size_t x_stride, int B, int H, int W,
                                                     int C, int KH, int KW, float *__restrict__ col) {
    const int K = KH * KW * C;
    const int pt = (KH - 1) / 2, pl = (KW - 1) / 2;
    const int npos = B * H * W;
    for (int pos = blockIdx.x; pos < npos; pos += gridDim.x) {
        const int w = pos % W, hb = pos / W, h = hb % H, b = hb / H;
        const float *xb = x + (size_t)b * x_stride;
        float *row = col + (size_t)pos * K;
        if ((C & 3) == 0) {
            const int C4 = C >> 2;
            for (int q = threadIdx.x; q < (K >> 2); q += 256) {
                const int tap = q / C4, c4 = q - tap * C4;
                const int dy = tap / KW, dx = tap - dy * KW;
                const int hh = h + dy - pt, ww = w + dx - pl;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                    v = *reinterpret_cast<const float4 *>(xb + ((size_t)hh * W + ww) * C + 4 * c4);
                *reinterpret_cast<float4 *>(row + 4 * q) = v;
            }
        } else {
            for (int kk = threadIdx.x; kk < K; kk += 256) {
                const int tap = kk / C, ci = kk - tap * C;
                const int dy = tap / KW, dx = tap - dy * KW;
                const int hh = h + dy - pt, ww = w + dx - pl;
                row[kk] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? xb[((size_t)hh * W + ww) * C + ci] : 0.f;
            }
        }
    }
}
// dX[b][h][w][ci] = sum over taps of dcol[(b, h - dy + pt, w - dx + pl)][(dy, dx, ci)]   (gather: no atomics).
// A workgroup takes 256 / min(C, 256) input positions at a time, lanes over channels, taps in a fixed order.
__global__ __launch_bounds__(256) void col2im_kernel(const float *__restrict__ dcol, int B, int H, int W, int C, int KH,
                                                     int KW, float *__restrict__ dx_) {
    const int K = KH * KW * C;
    const int pt = (KH - 1) / 2, pl = (KW - 1) / 2;
    const int npos = B * H * W;
    const int cw = min(C, 256), ppw = 256 / cw;            // channels per pass of a position, positions per workgroup
    const int sub = threadIdx.x / cw, c0 = threadIdx.x - sub * cw;
    for (int pos = blockIdx.x * ppw + sub; pos < npos && sub < ppw; pos += gridDim.x * ppw) {
        const int w = pos % W, hb = pos / W, h = hb % H, b = hb / H;
        for (int ci = c0; ci < C; ci += cw) {
            float s = 0.f;
            for (int dy = 0; dy < KH; ++dy) {
                const int ho = h - dy + pt;
                if (ho < 0 || ho >= H) continue;
                for (int dx = 0; dx < KW; ++dx) {
                    const int wo = w - dx + pl;
                    if (wo < 0 || wo >= W) continue;
                    s += dcol[(((size_t)b * H + ho) * W + wo) * K + (dy * KW + dx) * C + ci];
                }
            }
            dx_[(size_t)pos * C + ci] = s;
        }
    }
}


// ---- weight gradient of a KH x 16 convolution without the im2col matrix ------------------------------------------------
// dW[dy][dx][ci][co] = sum over (b, h, w) of x(b, h + dy - pt, w + dx - pl, ci) dZ(b, h, w, co).
// blockIdx.y = (dy, 32-channel block of ci, 32-channel block of co); a workgroup walks row segments (b, h, w0 .. w0 + seg) in
// a fixed stride, staging the ONE input row its dy needs (seg + 15 positions x 32 channels) and the dZ segment in LDS; wave v
// owns the four taps dx = 4 v .. 4 v + 3, i.e. four 32 x 32 accumulators on v_mfma_f32_32x32x2_f32 (A = x, rows = ci,
// k = position; B = dZ, columns = co), so one B fragment feeds four MFMAs.  Every workgroup writes its partial sums; they are
// added in workgroup order by splitk_reduce_kernel (no float atomics).
struct WgArgs {
    const float *x, *dz;
    float *part;
    int B, H, W, Cin, Cout, KH, pt, pl;
    int seg, nseg, ntiles;
};
template <int KW>
__global__ __launch_bounds__(256, 3) void wgrad_kernel(WgArgs a) {
    constexpr int TPW = KW / 4;                              // taps per wave
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    float *xs = wg_lds;                                      // [seg + KW - 1][32]
    float *ds = wg_lds + (size_t)(a.seg + KW - 1) * 32;      // [seg][32]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ncob = a.Cout / 32, ncib = a.Cin / 32;
    const int cob = blockIdx.y % ncob, cib = (blockIdx.y / ncob) % ncib, dy = blockIdx.y / (ncob * ncib);
    tf32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int half = lane >> 5, n = lane & 31;
    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int sgi = t % a.nseg, bh = t / a.nseg, h = bh % a.H, b = bh / a.H;
        const int hr = h + dy - a.pt;
        if (hr < 0 || hr >= a.H) continue;                   // the whole row is padding (uniform over the workgroup)
        const int w0 = sgi * a.seg, len = min(a.seg, a.W - w0);
        __syncthreads();
        const float *xrow = a.x + ((size_t)(b * a.H + hr) * a.W) * a.Cin + cib * 32;
        for (int idx = tid; idx < (a.seg + KW - 1) * 8; idx += 256) {
            const int pos = idx >> 3, c4 = idx & 7, ws = w0 - a.pl + pos;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ws >= 0 && ws < a.W) v = *reinterpret_cast<const float4 *>(xrow + (size_t)ws * a.Cin + c4 * 4);
            *reinterpret_cast<float4 *>(xs + pos * 32 + c4 * 4) = v;
        }
        const float *drow = a.dz + ((size_t)(b * a.H + h) * a.W + w0) * a.Cout + cob * 32;
        for (int idx = tid; idx < a.seg * 8; idx += 256) {
            const int pos = idx >> 3, c4 = idx & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < len) v = *reinterpret_cast<const float4 *>(drow + (size_t)pos * a.Cout + c4 * 4);
            *reinterpret_cast<float4 *>(ds + pos * 32 + c4 * 4) = v;
        }
        __syncthreads();
        const float *xa = xs + (half + wid * TPW) * 32 + n;
        const float *db = ds + half * 32 + n;
        const int ksteps = (len + 1) >> 1;
        for (int q = 0; q < ksteps; ++q) {
            const float bv = db[q * 64];
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[q * 64 + j * 32], bv, acc[j], 0, 0, 0);
        }
    }
    float *out = a.part + (size_t)blockIdx.x * ((size_t)a.KH * KW * a.Cin * a.Cout);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tap = dy * KW + wid * TPW + j;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = cib * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            out[((size_t)tap * a.Cin + ci) * a.Cout + cob * 32 + n] = acc[j][e];
        }
    }
}

// ---- column reductions over a [M][C] matrix: two stages, fixed order ---------------------------------
// mode 0 / 3: sum x;  1: sum (x - mu[c])^2;  2: (sum dy, sum dy * zhat) with zhat = (z - mu) * inv
#define CR_SPLIT 512                      // most row splits of a column reduction (the launcher picks 8 .. CR_SPLIT from M)
__global__ __launch_bounds__(256) void colreduce_kernel(const float *__restrict__ x, const float *__restrict__ z,
                                                         const float *__restrict__ mu, const float *__restrict__ inv,
                                                         size_t M, int C, int mode, int nsplit,
                                                         float *__restrict__ part0, float *__restrict__ part1) {
    // block = 8 row-lanes x 32 channels; blockIdx.x = channel block, blockIdx.y = row split
    __shared__ float red0[8][33], red1[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const size_t rows = (M + nsplit - 1) / nsplit;
    const size_t r0 = (size_t)blockIdx.y * rows, r1 = min(M, r0 + rows);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float m = (mode >= 1 && mu) ? mu[c] : 0.f, iv = mode == 2 ? inv[c] : 0.f;
#pragma unroll 4
        for (size_t r = r0 + rl; r < r1; r += 8) {
            const float v = x[r * C + c];
            if (mode == 0 || mode == 3) s0 += v;
            else if (mode == 1) { const float d = v - m; s0 += d * d; }
            else { s0 += v; s1 += v * ((z[r * C + c] - m) * iv); }
        }
    }
    red0[rl][cl] = s0; red1[rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < 8; ++i) { a += red0[i][cl]; b += red1[i][cl]; }
        part0[(size_t)blockIdx.y * C + c] = a;
        if (mode == 2) part1[(size_t)blockIdx.y * C + c] = b;
    }
}
// mode 0 -> out0 = sum / M (mean);  mode 1 -> out0 = var = sum / M, out1 = 1/sqrt(var + eps);
// mode 2 -> out0 = sum dy, out1 = sum dy zhat;  mode 3 -> out0 = plain sum
// One workgroup per 32 channels: eight row-lanes add nsplit / 8 partials each (fixed order), LDS combines them.
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float *__restrict__ part0, const float *__restrict__ part1,
                                                               int C, float invM, int mode, int nsplit,
                                                               float *__restrict__ out0, float *__restrict__ out1) {
    __shared__ float red0[8][33], red1[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
        int s = rl;
        for (; s + 56 < nsplit; s += 64) {                   // eight partials per lane in flight, added in order
            float q0[8], q1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                q0[u] = part0[(size_t)(s + 8 * u) * C + c];
                q1[u] = mode == 2 ? part1[(size_t)(s + 8 * u) * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += q0[u]; b += q1[u]; }
        }
        for (; s < nsplit; s += 8) { a += part0[(size_t)s * C + c]; if (mode == 2) b += part1[(size_t)s * C + c]; }
    }
    red0[rl][cl] = a; red1[rl][cl] = b;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    a = 0.f; b = 0.f;
    for (int i = 0; i < 8; ++i) { a += red0[i][cl]; b += red1[i][cl]; }
    if (mode == 0) out0[c] = a * invM;
    else if (mode == 1) { const float v = a * invM; out0[c] = v; out1[c] = 1.0f / sqrtf(v + TR_BN_EPS); }
    else if (mode == 2) { out0[c] = a; out1[c] = b; }
    else out0[c] = a;
}


// ---- fused BatchNormalization passes ---------------------------------------------------------------------------------
// For C a power of two (4 .. 1024; every BatchNormalization of the reference's graphs has 32 2^k channels).  All four
// kernels share one thread layout: a workgroup covers BNF_RPT * RL consecutive rows of the [M][C] matrix, RL = 1024 / C
// row lanes, one float4 (four channels) per thread and row.  Sums over row lanes are taken in lane order by one thread
// per channel quad, partial results per workgroup are combined in workgroup order by a second, tiny launch: fixed order,
// no float atomics.
//   bnf_stats_kernel        per-workgroup (mean, sum of squared deviations from that mean) of every channel; the final
//                           launch merges them pairwise (Chan et al.), so the variance never sees E[x^2] - mean^2
//   bnf_act_kernel          y = (z - mu) inv gamma + beta [-> sigmoid], plus max |output| per window for the next
//                           convolution's operand scaling
//   bnf_bwd_reduce_kernel   sum dy, sum dy zhat with dy = dA a (1 - a) formed on the fly (the sigmoid's gradient)
//   bnf_bwd_apply_kernel    dz = gamma inv (dy - mean dy - zhat mean(dy zhat)) [+ what is already in the destination],
//                           plus per-workgroup column sums of dz (the bias gradient of the convolution in front) and
//                           max |dz| per window (operand scaling of the data gradient)
#define BNF_RPT 8
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4_amax(float4 a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); }
// sum over the RL row lanes of channel quad cq, in lane order (called by the threads of row lane 0)
__device__ __forceinline__ float4 bnf_lane_sum(const float4 *red, int CQ, int RL, int cq) {
    float4 a = red[cq];
    for (int j = 1; j < RL; ++j) a = f4_add(a, red[j * CQ + cq]);
    return a;
}
// max |.| per window: rows r0 .. r0 + nrows of one workgroup usually lie in one window (HW rows each)
__device__ __forceinline__ void bnf_window_max(const float (&m)[BNF_RPT], int rl, int RL, size_t r0, int nrows, int HW,
                                               float *amax, float *lds16) {
    const size_t w0 = r0 / HW, w1 = (r0 + nrows - 1) / HW;
    for (size_t w = w0; w <= w1; ++w) {                      // one block reduction and ONE atomic per window touched
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < BNF_RPT; ++i) {
            const int r = rl + i * RL;
            if (w0 == w1 || (r < nrows && (r0 + r) / HW == w)) v = fmaxf(v, m[i]);
        }
        v = block_max(v, lds16);
        if (threadIdx.x == 0) atomicMax(reinterpret_cast<int *>(amax) + w, __float_as_int(v));
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void bnf_stats_kernel(const float *__restrict__ x, size_t M, int C,
                                                         float *__restrict__ pmean, float *__restrict__ pm2) {
    __shared__ float4 red[256];
    __shared__ float4 bc[256];
    const int CQ = C >> 2, RL = 256 / CQ, rl = threadIdx.x / CQ, cq = threadIdx.x - rl * CQ;
    const size_t r0 = (size_t)blockIdx.x * (RL * BNF_RPT);
    const int nrows = (int)min((size_t)(RL * BNF_RPT), M - r0);
    float4 v[BNF_RPT];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {
        const int r = rl + i * RL;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) { v[i] = *reinterpret_cast<const float4 *>(x + (r0 + r) * C + 4 * cq); s = f4_add(s, v[i]); }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        const float4 a = bnf_lane_sum(red, CQ, RL, cq);
        const float n = (float)nrows;
        bc[cq] = make_float4(a.x / n, a.y / n, a.z / n, a.w / n);
    }
    __syncthreads();
    const float4 mean = bc[cq];
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {
        if (rl + i * RL >= nrows) continue;
        const float dx = v[i].x - mean.x, dy = v[i].y - mean.y, dz = v[i].z - mean.z, dw = v[i].w - mean.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
    red[threadIdx.x] = q;
    __syncthreads();
    if (rl == 0) {
        const float4 a = bnf_lane_sum(red, CQ, RL, cq);
        *reinterpret_cast<float4 *>(pmean + (size_t)blockIdx.x * C + 4 * cq) = mean;
        *reinterpret_cast<float4 *>(pm2 + (size_t)blockIdx.x * C + 4 * cq) = a;
    }
}
__device__ __forceinline__ void bnf_merge(float &na, float &ma, float &qa, float nb, float mb, float qb) {
    if (nb <= 0.f) return;
    const float n = na + nb, d = mb - ma;
    ma += d * (nb / n);
    qa += qb + d * d * (na * nb / n);
    na = n;
}
// one workgroup per 32 channels; eight row lanes merge every eighth partial, lane 0 merges the eight results in order.
// Also the moving statistics (momentum 0.99, unbiased variance: see moving_update_kernel) when `update` is set.
__global__ __launch_bounds__(256) void bnf_stats_final_kernel(const float *__restrict__ pmean, const float *__restrict__ pm2,
                                                               size_t M, int C, int rows_wg, int nwg, float *__restrict__ mu,
                                                               float *__restrict__ var, float *__restrict__ inv,
                                                               float *__restrict__ mm, float *__restrict__ mv, int update,
                                                               float bessel) {
    __shared__ float sn[8][33], sm[8][33], sq[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float n = 0.f, m = 0.f, q = 0.f;
    if (c < C) {
        int s = rl;
        for (; s + 56 < nwg; s += 64) {                      // eight partials per lane in flight, merged in order
            float pm[8], pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { pm[u] = pmean[(size_t)(s + 8 * u) * C + c]; pq[u] = pm2[(size_t)(s + 8 * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                bnf_merge(n, m, q, (float)min((size_t)rows_wg, M - (size_t)(s + 8 * u) * rows_wg), pm[u], pq[u]);
        }
        for (; s < nwg; s += 8) {
            const float nb = (float)min((size_t)rows_wg, M - (size_t)s * rows_wg);
            bnf_merge(n, m, q, nb, pmean[(size_t)s * C + c], pm2[(size_t)s * C + c]);
        }
    }
    sn[rl][cl] = n; sm[rl][cl] = m; sq[rl][cl] = q;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    for (int i = 1; i < 8; ++i) bnf_merge(n, m, q, sn[i][cl], sm[i][cl], sq[i][cl]);
    const float v = q / (float)M;
    mu[c] = m; var[c] = v; inv[c] = 1.0f / sqrtf(v + TR_BN_EPS);
    if (update) {
        mm[c] = TR_BN_MOMENTUM * mm[c] + (1.0f - TR_BN_MOMENTUM) * m;
        mv[c] = TR_BN_MOMENTUM * mv[c] + (1.0f - TR_BN_MOMENTUM) * (v * bessel);
    }
}
__global__ __launch_bounds__(256) void bnf_act_kernel(const float *__restrict__ z, const float *__restrict__ mu,
                                                       const float *__restrict__ inv, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, size_t M, int C, int HW, int sig,
                                                       float *__restrict__ out, float *__restrict__ amax) {
    __shared__ float lds16[16];
    const int CQ = C >> 2, RL = 256 / CQ, rl = threadIdx.x / CQ, cq = threadIdx.x - rl * CQ;
    const size_t r0 = (size_t)blockIdx.x * (RL * BNF_RPT);
    const int nrows = (int)min((size_t)(RL * BNF_RPT), M - r0);
    const float4 mu4 = *reinterpret_cast<const float4 *>(mu + 4 * cq), iv4 = *reinterpret_cast<const float4 *>(inv + 4 * cq);
    const float4 g4 = *reinterpret_cast<const float4 *>(gamma + 4 * cq), b4 = *reinterpret_cast<const float4 *>(beta + 4 * cq);
    float m[BNF_RPT];
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {
        const int r = rl + i * RL;
        m[i] = 0.f;
        if (r >= nrows) continue;
        const float4 v = *reinterpret_cast<const float4 *>(z + (r0 + r) * C + 4 * cq);
        float4 y = make_float4((v.x - mu4.x) * iv4.x * g4.x + b4.x, (v.y - mu4.y) * iv4.y * g4.y + b4.y,
                               (v.z - mu4.z) * iv4.z * g4.z + b4.z, (v.w - mu4.w) * iv4.w * g4.w + b4.w);
        if (sig) y = make_float4(1.0f / (1.0f + expf(-y.x)), 1.0f / (1.0f + expf(-y.y)), 1.0f / (1.0f + expf(-y.z)),
                                 1.0f / (1.0f + expf(-y.w)));
        *reinterpret_cast<float4 *>(out + (r0 + r) * C + 4 * cq) = y;
        m[i] = f4_amax(y);
    }
    if (amax) bnf_window_max(m, rl, RL, r0, nrows, HW, amax, lds16);
}
__device__ __forceinline__ float4 bnf_dy(float4 d, const float *a, size_t off) {
    if (!a) return d;
    const float4 s = *reinterpret_cast<const float4 *>(a + off);
    return make_float4(d.x * s.x * (1.0f - s.x), d.y * s.y * (1.0f - s.y), d.z * s.z * (1.0f - s.z), d.w * s.w * (1.0f - s.w));
}
__global__ __launch_bounds__(256) void bnf_bwd_reduce_kernel(const float *__restrict__ dA, const float *__restrict__ a,
                                                              const float *__restrict__ z, const float *__restrict__ mu,
                                                              const float *__restrict__ inv, size_t M, int C,
                                                              float *__restrict__ part0, float *__restrict__ part1) {
    __shared__ float4 red[256];
    const int CQ = C >> 2, RL = 256 / CQ, rl = threadIdx.x / CQ, cq = threadIdx.x - rl * CQ;
    const size_t r0 = (size_t)blockIdx.x * (RL * BNF_RPT);
    const int nrows = (int)min((size_t)(RL * BNF_RPT), M - r0);
    const float4 mu4 = *reinterpret_cast<const float4 *>(mu + 4 * cq), iv4 = *reinterpret_cast<const float4 *>(inv + 4 * cq);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {
        const int r = rl + i * RL;
        if (r >= nrows) continue;
        const size_t off = (r0 + r) * C + 4 * cq;
        const float4 dy = bnf_dy(*reinterpret_cast<const float4 *>(dA + off), a, off);
        const float4 v = *reinterpret_cast<const float4 *>(z + off);
        s0 = f4_add(s0, dy);
        s1.x += dy.x * ((v.x - mu4.x) * iv4.x); s1.y += dy.y * ((v.y - mu4.y) * iv4.y);
        s1.z += dy.z * ((v.z - mu4.z) * iv4.z); s1.w += dy.w * ((v.w - mu4.w) * iv4.w);
    }
    red[threadIdx.x] = s0;
    __syncthreads();
    if (rl == 0) *reinterpret_cast<float4 *>(part0 + (size_t)blockIdx.x * C + 4 * cq) = bnf_lane_sum(red, CQ, RL, cq);
    __syncthreads();
    red[threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) *reinterpret_cast<float4 *>(part1 + (size_t)blockIdx.x * C + 4 * cq) = bnf_lane_sum(red, CQ, RL, cq);
}
__global__ __launch_bounds__(256) void bnf_bwd_apply_kernel(const float *__restrict__ dA, const float *__restrict__ a,
                                                             const float *__restrict__ z, const float *__restrict__ mu,
                                                             const float *__restrict__ inv, const float *__restrict__ gamma,
                                                             const float *__restrict__ sdy, const float *__restrict__ sdyz,
                                                             float invM, size_t M, int C, int HW, const float *acc, float *dst,
                                                             float *__restrict__ part2, float *__restrict__ gamax) {
    __shared__ float4 red[256];
    __shared__ float lds16[16];
    const int CQ = C >> 2, RL = 256 / CQ, rl = threadIdx.x / CQ, cq = threadIdx.x - rl * CQ;
    const size_t r0 = (size_t)blockIdx.x * (RL * BNF_RPT);
    const int nrows = (int)min((size_t)(RL * BNF_RPT), M - r0);
    const float4 mu4 = *reinterpret_cast<const float4 *>(mu + 4 * cq), iv4 = *reinterpret_cast<const float4 *>(inv + 4 * cq);
    const float4 g4 = *reinterpret_cast<const float4 *>(gamma + 4 * cq);
    const float4 a4 = *reinterpret_cast<const float4 *>(sdy + 4 * cq), b4 = *reinterpret_cast<const float4 *>(sdyz + 4 * cq);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float m[BNF_RPT];
#pragma unroll
    for (int i = 0; i < BNF_RPT; ++i) {
        const int r = rl + i * RL;
        m[i] = 0.f;
        if (r >= nrows) continue;
        const size_t off = (r0 + r) * C + 4 * cq;
        const float4 dy = bnf_dy(*reinterpret_cast<const float4 *>(dA + off), a, off);
        const float4 v = *reinterpret_cast<const float4 *>(z + off);
        float4 d;
        d.x = g4.x * iv4.x * (dy.x - a4.x * invM - ((v.x - mu4.x) * iv4.x) * (b4.x * invM));
        d.y = g4.y * iv4.y * (dy.y - a4.y * invM - ((v.y - mu4.y) * iv4.y) * (b4.y * invM));
        d.z = g4.z * iv4.z * (dy.z - a4.z * invM - ((v.z - mu4.z) * iv4.z) * (b4.z * invM));
        d.w = g4.w * iv4.w * (dy.w - a4.w * invM - ((v.w - mu4.w) * iv4.w) * (b4.w * invM));
        cs = f4_add(cs, d);
        m[i] = f4_amax(d);
        if (acc) d = f4_add(d, *reinterpret_cast<const float4 *>(acc + off));
        *reinterpret_cast<float4 *>(dst + off) = d;
    }
    if (part2) {
        red[threadIdx.x] = cs;
        __syncthreads();
        if (rl == 0) *reinterpret_cast<float4 *>(part2 + (size_t)blockIdx.x * C + 4 * cq) = bnf_lane_sum(red, CQ, RL, cq);
    }
    if (gamax) bnf_window_max(m, rl, RL, r0, nrows, HW, gamax, lds16);
}

// ---- elementwise ----------------------------------------------------------------------------------
__global__ void bn_apply_kernel(const float *__restrict__ z, const float *__restrict__ mu, const float *__restrict__ inv,
                                const float *__restrict__ gamma, const float *__restrict__ beta, size_t n, int C,
                                float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = (z[i] - mu[c]) * inv[c] * gamma[c] + beta[c];
    }
}
__global__ void bn_backward_kernel(const float *dy /* may alias dz */, const float *__restrict__ z, const float *__restrict__ mu,
                                   const float *__restrict__ inv, const float *__restrict__ gamma,
                                   const float *__restrict__ sdy, const float *__restrict__ sdyz, float invM, size_t n,
                                   int C, int training, float *dz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float zh = (z[i] - mu[c]) * inv[c];
        dz[i] = training ? gamma[c] * inv[c] * (dy[i] - sdy[c] * invM - zh * (sdyz[c] * invM)) : gamma[c] * inv[c] * dy[i];
    }
}
// The batch is normalised with the BIASED variance, but the moving variance absorbs the UNBIASED one: TensorFlow's
// fused batch norm hands var * M / (M - 1) to the running average (Keras 2.2's own path multiplies by
// M / (M - (1 + epsilon)): 1e-3 / M apart), M = the samples per channel of this batch.  `bessel` = M / (M - 1), 1 for M = 1.
__global__ void moving_update_kernel(float *__restrict__ mm, float *__restrict__ mv, const float *__restrict__ mu,
                                     const float *__restrict__ var, int C, float bessel) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mm[c] = TR_BN_MOMENTUM * mm[c] + (1.0f - TR_BN_MOMENTUM) * mu[c];
    mv[c] = TR_BN_MOMENTUM * mv[c] + (1.0f - TR_BN_MOMENTUM) * (var[c] * bessel);
}
__global__ void inv_from_var_kernel(const float *__restrict__ var, int C, float *__restrict__ inv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) inv[c] = 1.0f / sqrtf(var[c] + TR_BN_EPS);
}
__global__ void sigmoid_fwd_kernel(const float *__restrict__ x, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float *dy, const float *__restrict__ a, size_t n, float *dx) {   // dy may alias dx
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * a[i] * (1.0f - a[i]);
}
__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n, float *__restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a[i] + b[i];
}
__global__ void accumulate_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}
// pooling (valid, stride = pool); out rows may be written with a row pitch (flatten into the dense input)
__global__ void pool_fwd_kernel(const float *__restrict__ x, int B, int H, int W, int C, int PH, int PW, int is_max,
                                float *__restrict__ y, size_t y_stride) {
    const int HO = H / PH, WO = W / PW;
    const size_t total = (size_t)B * HO * WO * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int wo = (int)(r % WO); r /= WO;
        const int ho = (int)(r % HO);
        const int b = (int)(r / HO);
        float m = is_max ? -INFINITY : 0.f;
        for (int dy = 0; dy < PH; ++dy)
            for (int dx = 0; dx < PW; ++dx) {
                const float v = x[(((size_t)b * H + ho * PH + dy) * W + wo * PW + dx) * C + c];
                m = is_max ? fmaxf(m, v) : m + v;
            }
        if (!is_max) m /= (float)(PH * PW);
        y[(size_t)b * y_stride + ((size_t)ho * WO + wo) * C + c] = m;
    }
}
// gradient of the pooling: max routes to the FIRST maximum in (dy, dx) order; the part of x the valid
// pooling never read gets zero
__global__ void pool_bwd_kernel(const float *__restrict__ dy_, size_t dy_stride, const float *__restrict__ x, int B, int H,
                                int W, int C, int PH, int PW, int is_max, float *__restrict__ dx_) {
    const int HO = H / PH, WO = W / PW;
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int b = (int)(r / H);
        const int ho = h / PH, wo = w / PW;
        float g = 0.f;
        if (ho < HO && wo < WO) {
            const float d = dy_[(size_t)b * dy_stride + ((size_t)ho * WO + wo) * C + c];
            if (!is_max) g = d / (float)(PH * PW);
            else {
                int best = 0; float bv = -INFINITY;
                for (int dy = 0; dy < PH; ++dy)
                    for (int dx = 0; dx < PW; ++dx) {
                        const float v = x[(((size_t)b * H + ho * PH + dy) * W + wo * PW + dx) * C + c];
                        if (v > bv) { bv = v; best = dy * PW + dx; }
                    }
                g = ((h - ho * PH) * PW + (w - wo * PW)) == best ? d : 0.f;
            }
        }
        dx_[i] = g;
    }
}
__global__ void copy_rows_kernel(const float *__restrict__ src, size_t src_stride, float *__restrict__ dst,
                                 size_t dst_stride, int B, size_t n) {
    const size_t total = (size_t)B * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / n, j = i - b * n;
        dst[b * dst_stride + j] = src[b * src_stride + j];
    }
}
// loss + gradient wrt the last Dense's output.  K == 1: pred = sigmoid(z), loss = mean (pred - y)^2 (Keras
// mean_squared_error on the activation-scale target); K > 1: softmax + sparse categorical cross-entropy.
__global__ void loss_kernel(const float *__restrict__ z, const float *__restrict__ y, int B, int K, float *__restrict__ pred,
                            float *__restrict__ dz, float *__restrict__ loss_rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *l = z + (size_t)b * K;
    if (K == 1) {
        const float p = 1.0f / (1.0f + expf(-l[0]));
        const float d = p - y[b];
        pred[b] = p;
        loss_rows[b] = d * d;
        dz[b] = 2.0f * d / (float)B * p * (1.0f - p);
    } else {
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, l[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(l[k] - m);
        int yi = (int)y[b];
        yi = yi < 0 ? 0 : (yi >= K ? K - 1 : yi);
        for (int k = 0; k < K; ++k) {
            const float p = expf(l[k] - m) / s;
            pred[(size_t)b * K + k] = p;
            dz[(size_t)b * K + k] = (p - (k == yi ? 1.0f : 0.0f)) / (float)B;
        }
        loss_rows[b] = -(l[yi] - m - logf(s));
    }
}
__global__ void adagrad_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ acc, size_t n,
                               float lr, float eps) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float a = acc[i] + gi * gi;
        acc[i] = a;
        p[i] = p[i] - lr * gi / (sqrtf(a) + eps);
    }
}

// =====================================================================================
// host side
// =====================================================================================
namespace {

struct Param { float *w = nullptr, *g = nullptr, *acc = nullptr; size_t n = 0; bool trainable = true; };

enum OpKind { OP_CONV, OP_BN, OP_SIGMOID, OP_ADD, OP_POOL, OP_FLATTEN, OP_DENSE };
struct Op {
    OpKind kind;
    int in0 = -1, in1 = -1, out = -1;             // tensor ids
    int H = 0, W = 0, Cin = 0, Cout = 0, kh = 1, kw = 1;    // conv / dense (H = W = 1) / pool (kh, kw = pool)
    int is_max = 0;
    int p0 = -1, p1 = -1, p2 = -1, p3 = -1;       // params: conv/dense (kernel, bias); bn (gamma, beta, mean, var)
    int flat_off = 0;                              // OP_FLATTEN: column offset in the dense input
    float *bmu = nullptr, *binv = nullptr, *bvar = nullptr;      // BN batch statistics of the last forward
    // OP_CONV on the split-fp16 kernels (amt_convh.h): forward and, when the transposed shape is covered too, the data
    // gradient; job = index of the layer's weight-preparation job (-1: the GEMM path)
    int job = -1;
    bool fast_bwd = false;
    amt_convh_plan fplan, bplan;
    void *wp_f = nullptr, *wp_b = nullptr;
    // OP_BN on the fused passes (bnf_*): sig_out >= 0 when the sigmoid that follows is folded in (its output tensor is
    // written directly and the OP_SIGMOID is skipped, forwards and backwards)
    bool bn_fused = false;
    int sig_out = -1;
    bool skip = false;                             // OP_SIGMOID folded into the BatchNormalization in front of it
};
// g is the tensor's own gradient buffer; gptr is where its gradient currently lives (its own buffer, or -- after a
// shortcut Add handed the same gradient to two tensors -- the buffer of the tensor it came from: gradients are only
// ever written into the destination's own buffer, never in place into a source, so such a view stays valid).
// am / gam: per-window max |v| / max |g| left by the producing kernel this step (null: not measured)
struct Tensor {
    int H, W, C; bool flat = false;
    float *v = nullptr, *g = nullptr, *gptr = nullptr; bool g_set = false;
    float *am_slot = nullptr, *gam_slot = nullptr, *am = nullptr, *gam = nullptr;
};

static unsigned grid1(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 65535); }

}  // namespace

struct amt_trainer {
    amt_rdcnn_desc d;
    std::vector<Param> params;                    // canonical order (rdcnn.py pack_weights)
    std::vector<Op> ops;
    std::vector<Tensor> tensors;
    std::vector<int> inputs;                      // tensor id of each tower's input
    int t_flat = -1, t_logits = -1;
    int flat = 0, capB = 0;
    float lr = 0.01f, eps = 1e-7f, acc0 = 0.f;
    float *col = nullptr, *part = nullptr, *red0 = nullptr, *red1 = nullptr, *stat0 = nullptr, *stat1 = nullptr;
    float *pred = nullptr, *loss_rows = nullptr, *dlogits = nullptr;
    size_t col_cap = 0, part_cap = 0;
    std::vector<float *> allocs;
    // split-fp16 convolutions: per-layer weight jobs (device copy), max |w| / exponent slots, per-window operand maxima
    std::vector<amt_convh_pack_job> jobs;
    amt_convh_pack_job *jobs_dev = nullptr;
    float *wmax = nullptr;
    int *sw = nullptr;
    float *amax = nullptr;                        // [2 jobs][capB] scratch + [2 tensors][capB] producer-measured maxima
    size_t amax_floats = 0;
    int max_job_elems = 0;
    // parameters live in three arenas of one layout (weights, gradients, Adagrad accumulators): one update launch
    float *w_arena = nullptr, *g_arena = nullptr, *a_arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    float *red2 = nullptr;                        // column sums of dz per workgroup (bias gradient of the convolution in front)
    size_t red_cap = 0;
    size_t fast_min_m = 2048;                     // AMT_TRAIN_FAST_MIN_M at create: fewest output positions (B H W) for the fast forms
    bool wgrad_direct_on = true;                  // AMT_TRAIN_WGRAD=0 at create: weight gradients through im2col + GEMM
    int colsum_of = -1;                           // tensor whose gradient's column partials red2 holds (-1: none)
    int colsum_nwg = 0;
};

namespace {

int talloc(amt_trainer *t, size_t n, float **out, bool zero = false) {
    float *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)) != hipSuccess) return AMT_E_NOMEM;
    if (zero && hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(float)) != hipSuccess) return AMT_E_HIP;
    t->allocs.push_back(p);
    *out = p;
    return AMT_OK;
}

int add_param(amt_trainer *t, const float *&cur, size_t n, bool trainable) {
    Param p;
    p.n = n; p.trainable = trainable;
    const size_t off = t->arena_used;
    if (off + n > t->arena_cap) return -1;
    t->arena_used = (off + n + 63) & ~(size_t)63;                 // 256-byte aligned slices
    p.w = t->w_arena + off;
    if (hipMemcpy(p.w, cur, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -1;
    cur += n;
    if (trainable) { p.g = t->g_arena + off; p.acc = t->a_arena + off; }
    t->params.push_back(p);
    return (int)t->params.size() - 1;
}

int new_tensor(amt_trainer *t, int H, int W, int C, bool flat = false) {
    Tensor x; x.H = H; x.W = W; x.C = C; x.flat = flat;
    t->tensors.push_back(x);
    return (int)t->tensors.size() - 1;
}

// GEMM dispatch with split over the contraction when the output grid is small
int gemm(amt_trainer *t, bool TA, bool TB, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
         int M, int N, int K, const float *bias, hipStream_t st) {
    const int wn = N <= 32 ? 1 : 2;
    const int tm = 32 * (4 / wn), tn = 32 * wn;
    const int gx = (N + tn - 1) / tn, gy = (M + tm - 1) / tm;
    int Z = 1;
    const long tiles = (long)gx * gy;
    if (tiles < 512 && K > 4 * TG_KC) Z = (int)std::min<long>((512 + tiles - 1) / tiles, (K + 4 * TG_KC - 1) / (4 * TG_KC));
    Z = std::min(Z, 256);
    int kslice = ((K + Z - 1) / Z + TG_KC - 1) / TG_KC * TG_KC;
    Z = (K + kslice - 1) / kslice;
    float *part = nullptr;
    if (Z > 1) {
        const size_t need = (size_t)Z * M * N;
        if (need > t->part_cap) {
            if (talloc(t, need, &t->part) != AMT_OK) return AMT_E_NOMEM;
            t->part_cap = need;
        }
        part = t->part;
    }
    dim3 grid(gx, gy, Z);
#define TG_LAUNCH(ta, tb)                                                                                       \
    do {                                                                                                        \
        if (wn == 1) tgemm_kernel<ta, tb, 1><<<grid, 256, 0, st>>>(A, lda, B, ldb, C, ldc, M, N, K, kslice, bias, part); \
        else tgemm_kernel<ta, tb, 2><<<grid, 256, 0, st>>>(A, lda, B, ldb, C, ldc, M, N, K, kslice, bias, part);  \
    } while (0)
    if (TA && TB) return AMT_E_UNSUPPORTED;
    if (TA) TG_LAUNCH(true, false); else if (TB) TG_LAUNCH(false, true); else TG_LAUNCH(false, false);
#undef TG_LAUNCH
    if (Z > 1) splitk_reduce_kernel<<<grid1((size_t)M * N), 256, 0, st>>>(part, Z, C, ldc, M, N, bias);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int colreduce(amt_trainer *t, const float *x, const float *z, const float *mu, const float *inv, size_t M, int C, int mode,
              float *out0, float *out1, hipStream_t st) {
    // ~64 rows per row-lane and split, between 8 splits and enough workgroups to cover the chip twice
    int nsplit = (int)std::min<size_t>(CR_SPLIT, std::max<size_t>(8, M / 512));
    nsplit = (nsplit + 7) & ~7;
    colreduce_kernel<<<dim3((C + 31) / 32, nsplit), 256, 0, st>>>(x, z, mu, inv, M, C, mode, nsplit, t->red0, t->red1);
    colreduce_final_kernel<<<(C + 31) / 32, 256, 0, st>>>(t->red0, t->red1, C, 1.0f / (float)M, mode, nsplit, out0, out1);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// weight gradient without im2col (wgrad_kernel): KH x 16 kernels on 32-multiples of channels
bool wgrad_direct_ok(const Op &o) { return o.kw == 16 && o.Cin % 32 == 0 && o.Cout % 32 == 0 && o.W >= 16; }
int wgrad_direct(amt_trainer *t, const Op &o, const float *x, const float *dz, float *dW, int B, hipStream_t st) {
    WgArgs a;
    a.x = x; a.dz = dz; a.B = B; a.H = o.H; a.W = o.W; a.Cin = o.Cin; a.Cout = o.Cout; a.KH = o.kh;
    a.pt = (o.kh - 1) / 2; a.pl = (o.kw - 1) / 2;
    a.nseg = (o.W + 191) / 192;
    a.seg = ((o.W + a.nseg - 1) / a.nseg + 1) & ~1;
    a.ntiles = B * o.H * a.nseg;
    const int ngroups = o.kh * (o.Cin / 32) * (o.Cout / 32);
    const int P = std::max(1, std::min(a.ntiles, 512 / ngroups));
    const size_t nW = (size_t)o.kh * o.kw * o.Cin * o.Cout, need = (size_t)P * nW;
    if (need > t->part_cap) {
        if (talloc(t, need, &t->part) != AMT_OK) return AMT_E_NOMEM;
        t->part_cap = need;
    }
    a.part = t->part;
    const size_t lds = (size_t)(2 * a.seg + 15) * 32 * sizeof(float);
    wgrad_kernel<16><<<dim3(P, ngroups), 256, lds, st>>>(a);
    splitk_reduce_kernel<<<grid1(nW), 256, 0, st>>>(t->part, P, dW, o.Cout, o.kh * o.kw * o.Cin, o.Cout, nullptr);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

}  // namespace

extern "C" {

int amt_trainer_destroy(amt_trainer *t) {
    if (!t) return AMT_OK;
    for (float *p : t->allocs) (void)hipFree(p);
    delete t;
    return AMT_OK;
}

int amt_trainer_create(amt_trainer **out, const amt_rdcnn_desc *desc, const float *wh, size_t n_floats, float lr,
                       float epsilon, float initial_accumulator) {
    if (!out || !desc || !wh) return AMT_E_INVALID;
    if ((size_t)amt_rdcnn_param_count(desc) != n_floats || n_floats == 0) return AMT_E_SHAPE;
    const amt_rdcnn_desc &d = *desc;
    amt_trainer *t = new amt_trainer();
    t->d = d;
    if (lr > 0.f) t->lr = lr;
    if (epsilon > 0.f) t->eps = epsilon;
    t->acc0 = initial_accumulator > 0.f ? initial_accumulator : 0.f;
    {
        // upper bound of the slices' padding: 8 parameters per convolution (kernel, bias, BN x 2 incl. shortcut) + the head
        const size_t nparams = (size_t)d.n_towers * (size_t)d.conv_layers * 16 + 64;
        t->arena_cap = n_floats + 64 * nparams;
        if (talloc(t, t->arena_cap, &t->w_arena, true) != AMT_OK || talloc(t, t->arena_cap, &t->g_arena, true) != AMT_OK ||
            talloc(t, t->arena_cap, &t->a_arena, true) != AMT_OK) {
            amt_trainer_destroy(t); return AMT_E_NOMEM;
        }
        if (t->acc0 > 0.f) {
            std::vector<float> a0(t->arena_cap, t->acc0);
            if (hipMemcpy(t->a_arena, a0.data(), a0.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                amt_trainer_destroy(t); return AMT_E_HIP;
            }
        }
    }
    const float *cur = wh;
#define TR_P(n, tr) add_param(t, cur, (size_t)(n), tr)
#define TR_FAIL() do { amt_trainer_destroy(t); return AMT_E_NOMEM; } while (0)
    auto bn_op = [&](int in, int C) -> int {
        Op o; o.kind = OP_BN; o.in0 = in; o.Cout = C;
        o.p0 = TR_P(C, true); o.p1 = TR_P(C, true); o.p2 = TR_P(C, false); o.p3 = TR_P(C, false);
        if (o.p0 < 0 || o.p1 < 0 || o.p2 < 0 || o.p3 < 0) return -1;
        const Tensor &x = t->tensors[in];
        o.out = new_tensor(t, x.H, x.W, C);
        if (talloc(t, C, &o.bmu) != AMT_OK || talloc(t, C, &o.binv) != AMT_OK || talloc(t, C, &o.bvar) != AMT_OK) return -1;
        t->ops.push_back(o);
        return o.out;
    };
    std::vector<std::pair<int, int>> tails;               // (tensor, flat offset)
    int flat = 0;
    for (int tw = 0; tw < d.n_towers; ++tw) {
        int H = d.in_h[tw], W = d.in_w[tw], C = 1, fo = 32;
        int cur_t = new_tensor(t, H, W, 1);
        t->inputs.push_back(cur_t);
        int p0 = cur_t;
        for (int i = 1; i <= d.conv_layers; ++i) {
            Op c; c.kind = OP_CONV; c.in0 = cur_t; c.H = H; c.W = W; c.Cin = C; c.Cout = fo; c.kh = d.kh[tw]; c.kw = d.kw[tw];
            c.p0 = TR_P((size_t)c.kh * c.kw * C * fo, true); c.p1 = TR_P(fo, true);
            if (c.p0 < 0 || c.p1 < 0) TR_FAIL();
            c.out = new_tensor(t, H, W, fo);
            t->ops.push_back(c);
            int z = bn_op(c.out, fo);
            if (z < 0) TR_FAIL();
            Op s; s.kind = OP_SIGMOID; s.in0 = z; s.out = new_tensor(t, H, W, fo);
            t->ops.push_back(s);
            cur_t = s.out; C = fo;
            if (d.residual_frequency > 0 && i % d.residual_frequency == 0) {
                int a = p0;
                const Tensor src = t->tensors[p0];
                if (!(src.H == H && src.W == W && src.C == C)) {
                    if (src.C != C) {
                        Op pc; pc.kind = OP_CONV; pc.in0 = a; pc.H = src.H; pc.W = src.W; pc.Cin = src.C; pc.Cout = C; pc.kh = 1; pc.kw = 1;
                        pc.p0 = TR_P((size_t)src.C * C, true); pc.p1 = TR_P(C, true);
                        if (pc.p0 < 0 || pc.p1 < 0) TR_FAIL();
                        pc.out = new_tensor(t, src.H, src.W, C);
                        t->ops.push_back(pc);
                        a = pc.out;
                    }
                    if (src.H != H || src.W != W) {
                        Op ap; ap.kind = OP_POOL; ap.in0 = a; ap.is_max = 0; ap.kh = src.H / H; ap.kw = src.W / W;
                        ap.H = src.H; ap.W = src.W; ap.Cin = C;
                        if (src.H / ap.kh != H || src.W / ap.kw != W) { amt_trainer_destroy(t); return AMT_E_UNSUPPORTED; }
                        ap.out = new_tensor(t, H, W, C);
                        t->ops.push_back(ap);
                        a = ap.out;
                    }
                    a = bn_op(a, C);
                    if (a < 0) TR_FAIL();
                }
                Op ad; ad.kind = OP_ADD; ad.in0 = a; ad.in1 = cur_t; ad.out = new_tensor(t, H, W, C);
                t->ops.push_back(ad);
                cur_t = bn_op(ad.out, C);
                if (cur_t < 0) TR_FAIL();
                p0 = cur_t;
            }
            if (d.pool_layer_frequency > 0 && i % d.pool_layer_frequency == 0) {
                Op mp; mp.kind = OP_POOL; mp.in0 = cur_t; mp.is_max = 1; mp.kh = d.pool_h[tw]; mp.kw = d.pool_w[tw];
                mp.H = H; mp.W = W; mp.Cin = C;
                H /= mp.kh; W /= mp.kw;
                if (H < 1 || W < 1) { amt_trainer_destroy(t); return AMT_E_UNSUPPORTED; }
                mp.out = new_tensor(t, H, W, C);
                t->ops.push_back(mp);
                cur_t = mp.out;
            }
            if (d.feature_expand_frequency > 0 && i % d.feature_expand_frequency == 0) fo *= 2;
        }
        tails.push_back({cur_t, flat});
        flat += H * W * C;
    }
    t->flat = flat;
    t->t_flat = new_tensor(t, 1, 1, flat, true);
    for (auto &tl : tails) {
        Op f; f.kind = OP_FLATTEN; f.in0 = tl.first; f.out = t->t_flat; f.flat_off = tl.second;
        t->ops.push_back(f);
    }
    {
        Op d1; d1.kind = OP_DENSE; d1.in0 = t->t_flat; d1.Cin = flat; d1.Cout = d.dense_units;
        d1.p0 = TR_P((size_t)flat * d.dense_units, true); d1.p1 = TR_P(d.dense_units, true);
        if (d1.p0 < 0 || d1.p1 < 0) TR_FAIL();
        d1.out = new_tensor(t, 1, 1, d.dense_units, true);
        t->ops.push_back(d1);
        Op s; s.kind = OP_SIGMOID; s.in0 = d1.out; s.out = new_tensor(t, 1, 1, d.dense_units, true);
        t->ops.push_back(s);
        Op d2; d2.kind = OP_DENSE; d2.in0 = s.out; d2.Cin = d.dense_units; d2.Cout = d.output_classes;
        d2.p0 = TR_P((size_t)d.dense_units * d.output_classes, true); d2.p1 = TR_P(d.output_classes, true);
        if (d2.p0 < 0 || d2.p1 < 0) TR_FAIL();
        d2.out = new_tensor(t, 1, 1, d.output_classes, true);
        t->ops.push_back(d2);
        t->t_logits = d2.out;
    }
#undef TR_P
#undef TR_FAIL
    if ((size_t)(cur - wh) != n_floats) { amt_trainer_destroy(t); return AMT_E_SHAPE; }
    {
        const char *e = getenv("AMT_TRAIN_WGRAD");
        t->wgrad_direct_on = !(e && atoi(e) == 0);
        const char *m = getenv("AMT_TRAIN_FAST_MIN_M");
        if (m && atol(m) >= 0) t->fast_min_m = (size_t)atol(m);
    }
    // BatchNormalizations on the fused passes, with the sigmoid behind them folded in (AMT_TRAIN_FUSED_BN=0: the generic
    // column-reduction kernels)
    {
        const char *e = getenv("AMT_TRAIN_FUSED_BN");
        const bool want = !(e && atoi(e) == 0);
        for (size_t i = 0; i < t->ops.size(); ++i) {
            Op &o = t->ops[i];
            if (!want || o.kind != OP_BN) continue;
            const int C = o.Cout;
            if (C < 4 || C > 1024 || (C & (C - 1)) != 0) continue;
            o.bn_fused = true;
            if (i + 1 < t->ops.size() && t->ops[i + 1].kind == OP_SIGMOID && t->ops[i + 1].in0 == o.out) {
                o.sig_out = t->ops[i + 1].out;
                t->ops[i + 1].skip = true;
            }
        }
    }
    // convolutions the split-fp16 kernels cover (AMT_TRAIN_FAST=0: keep every layer on the f32 GEMM path)
    {
        const char *e = getenv("AMT_TRAIN_FAST");
        const bool want = !(e && atoi(e) == 0);
        int njobs = 0;
        for (Op &o : t->ops) {
            if (!want || o.kind != OP_CONV || (o.kh == 1 && o.kw == 1)) continue;
            if (amt_convh_plan_init(&o.fplan, o.kh, o.kw, o.Cin, o.Cout, o.H, o.W) != AMT_OK) continue;
            o.job = njobs++;
            const bool is_input = std::find(t->inputs.begin(), t->inputs.end(), o.in0) != t->inputs.end();
            o.fast_bwd = !is_input && amt_convh_plan_init(&o.bplan, o.kh, o.kw, o.Cout, o.Cin, o.H, o.W) == AMT_OK;
        }
        if (njobs) {
            float *tmp = nullptr;
            if (talloc(t, njobs, &t->wmax, true) != AMT_OK || talloc(t, njobs, &tmp, true) != AMT_OK) {
                amt_trainer_destroy(t); return AMT_E_NOMEM;
            }
            t->sw = reinterpret_cast<int *>(tmp);
            t->jobs.resize(njobs);
            for (Op &o : t->ops) {
                if (o.job < 0) continue;
                float *a = nullptr, *b = nullptr;
                if (talloc(t, (amt_convh_packed_bytes(&o.fplan) + 3) / 4, &a) != AMT_OK ||
                    (o.fast_bwd && talloc(t, (amt_convh_packed_bytes(&o.bplan) + 3) / 4, &b) != AMT_OK)) {
                    amt_trainer_destroy(t); return AMT_E_NOMEM;
                }
                o.wp_f = a; o.wp_b = b;
                amt_convh_pack_job &j = t->jobs[o.job];
                j.w = t->params[o.p0].w; j.packed_fwd = a; j.packed_bwd = b;
                j.wmax = t->wmax + o.job; j.sw = t->sw + o.job;
                j.ntap = o.kh * o.kw; j.Cin = o.Cin; j.Cout = o.Cout;
                t->max_job_elems = std::max(t->max_job_elems, j.ntap * j.Cin * j.Cout);
            }
            float *jd = nullptr;
            const size_t jb = (size_t)njobs * sizeof(amt_convh_pack_job);
            if (talloc(t, (jb + 3) / 4, &jd) != AMT_OK ||
                hipMemcpy(jd, t->jobs.data(), jb, hipMemcpyHostToDevice) != hipSuccess) {
                amt_trainer_destroy(t); return AMT_E_NOMEM;
            }
            t->jobs_dev = reinterpret_cast<amt_convh_pack_job *>(jd);
        }
    }
    int maxC = 1;
    for (const Tensor &x : t->tensors) maxC = std::max(maxC, x.C);
    if (talloc(t, maxC, &t->stat0) != AMT_OK || talloc(t, maxC, &t->stat1) != AMT_OK) {
        amt_trainer_destroy(t); return AMT_E_NOMEM;
    }
    *out = t;
    return AMT_OK;
}

static size_t bnf_rows_wg(int C) { return (size_t)(1024 / C) * BNF_RPT; }
static int bnf_nwg(size_t M, int C) { return (int)((M + bnf_rows_wg(C) - 1) / bnf_rows_wg(C)); }

static int ensure_batch(amt_trainer *t, int B) {
    if (B <= t->capB) return AMT_OK;
    // (re)allocate activations and gradients for B windows; earlier, smaller buffers stay owned until destroy
    size_t col_need = 0;
    for (Tensor &x : t->tensors) {
        const size_t n = (size_t)B * x.H * x.W * x.C;
        if (talloc(t, n, &x.v) != AMT_OK || talloc(t, n, &x.g) != AMT_OK) return AMT_E_NOMEM;
    }
    int maxC = 1;
    size_t red_need = 0;
    for (const Op &o : t->ops) {
        if (o.kind == OP_CONV) col_need = std::max(col_need, (size_t)B * o.H * o.W * o.kh * o.kw * o.Cin);
        if (o.kind == OP_BN) {
            const Tensor &in = t->tensors[o.in0];
            maxC = std::max(maxC, o.Cout);
            if (o.bn_fused) red_need = std::max(red_need, (size_t)bnf_nwg((size_t)B * in.H * in.W, o.Cout) * o.Cout);
        }
        if (o.kind == OP_DENSE) { maxC = std::max(maxC, o.Cout); col_need = std::max(col_need, (size_t)B * o.Cin); }
    }
    for (const Tensor &x : t->tensors) col_need = std::max(col_need, (size_t)B * x.H * x.W * x.C);
    red_need = std::max(red_need, (size_t)CR_SPLIT * maxC);
    if (talloc(t, col_need, &t->col) != AMT_OK) return AMT_E_NOMEM;
    t->col_cap = col_need;
    if (talloc(t, red_need, &t->red0) != AMT_OK || talloc(t, red_need, &t->red1) != AMT_OK ||
        talloc(t, red_need, &t->red2) != AMT_OK)
        return AMT_E_NOMEM;
    t->red_cap = red_need;
    // per-window maxima: two scratch rows per split-fp16 layer (operands no producer measured) + two per tensor
    t->amax_floats = (2 * t->jobs.size() + 2 * t->tensors.size()) * (size_t)B;
    if (talloc(t, t->amax_floats, &t->amax) != AMT_OK) return AMT_E_NOMEM;
    {
        float *base = t->amax + 2 * t->jobs.size() * (size_t)B;
        for (Tensor &x : t->tensors) { x.am_slot = base; x.gam_slot = base + B; base += 2 * (size_t)B; }
    }
    const int K = t->d.output_classes;
    if (talloc(t, (size_t)B * K, &t->pred) != AMT_OK || talloc(t, B, &t->loss_rows) != AMT_OK ||
        talloc(t, (size_t)B * K, &t->dlogits) != AMT_OK)
        return AMT_E_NOMEM;
    t->capB = B;
    return AMT_OK;
}

// ---- gradient hand-over ------------------------------------------------------------------------------------------------
// A kernel that can write "value (+ what is there)" asks claim() for the destination's own buffer and the addend (null on
// the first arrival); afterwards the tensor's gradient lives in its own buffer.
static float *claim(amt_trainer *t, int id, const float **acc) {
    Tensor &x = t->tensors[id];
    *acc = x.g_set ? x.gptr : nullptr;
    x.gptr = x.g; x.g_set = true; x.gam = nullptr;
    return x.g;
}
// the gradient of tensor `id` is (also) the n floats at src: a view on the first arrival, a sum into the own buffer after
static int give_view(amt_trainer *t, int id, const float *src, size_t n, hipStream_t st) {
    Tensor &x = t->tensors[id];
    if (!x.g_set) { x.gptr = const_cast<float *>(src); x.g_set = true; x.gam = nullptr; return AMT_OK; }
    add_kernel<<<grid1(n), 256, 0, st>>>(x.gptr, src, n, x.g);
    x.gptr = x.g; x.gam = nullptr;
    return AMT_OK;
}
// src is a scratch buffer that will be reused: copy (first arrival) or add into the own buffer
static int give_copy(amt_trainer *t, int id, const float *src, size_t n, hipStream_t st) {
    Tensor &x = t->tensors[id];
    if (!x.g_set) {
        if (x.g != src) AMT_HIP_CHECK(hipMemcpyAsync(x.g, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        add_kernel<<<grid1(n), 256, 0, st>>>(x.gptr, src, n, x.g);
    }
    x.gptr = x.g; x.g_set = true; x.gam = nullptr;
    return AMT_OK;
}

int amt_trainer_step(amt_trainer *t, const float *const *x, const float *y, int B, int update, float *loss_host,
                     float *pred_out, void *stream) {
    if (!t || !x || !y || B <= 0) return AMT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_batch(t, B);
    if (rc != AMT_OK) return rc;
    const int training = update ? 1 : 0;
    const int K = t->d.output_classes;
    for (size_t i = 0; i < t->inputs.size(); ++i) {
        if (!x[i]) return AMT_E_INVALID;
        Tensor &in = t->tensors[t->inputs[i]];
        AMT_HIP_CHECK(hipMemcpyAsync(in.v, x[i], (size_t)B * in.H * in.W * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    AMT_HIP_CHECK(hipMemsetAsync(t->amax, 0, t->amax_floats * sizeof(float), st));
    for (Tensor &x_ : t->tensors) { x_.am = nullptr; x_.gam = nullptr; x_.g_set = false; x_.gptr = nullptr; }
    t->colsum_of = -1;
    if (!t->jobs.empty()) {
        // the weights moved in the last update: measure their range and lay them out for the split-fp16 kernels again
        // (two launches for every layer)
        const int nj = (int)t->jobs.size();
        AMT_HIP_CHECK(hipMemsetAsync(t->wmax, 0, nj * sizeof(float), st));
        rc = amt_convh_pack_all(t->jobs_dev, nj, t->max_job_elems, st);
        if (rc != AMT_OK) return rc;
    }
    // layers too small to fill the chip stay on the GEMM path (a handful of workgroups walking the whole contraction
    // one after the other is slower than the split-K GEMM)
    auto fast_now = [&](const Op &o) { return o.job >= 0 && (size_t)B * o.H * o.W >= t->fast_min_m; };
    // AMT_TRAIN_TRACE=1: synchronise and report after every op (locating a faulting kernel)
    static const bool trace = getenv("AMT_TRAIN_TRACE") != nullptr;
    int op_no = 0;
    auto mark = [&](const char *phase, const Op &o) {
        if (!trace) return;
        const hipError_t e = hipStreamSynchronize(st);
        fprintf(stderr, "[amt_train] %s op %d kind %d H %d W %d Cin %d Cout %d k %dx%d : %s\n", phase, op_no, (int)o.kind, o.H, o.W,
                o.Cin, o.Cout, o.kh, o.kw, hipGetErrorString(e));
        fflush(stderr);
    };
    // ---------------- forward ----------------------------------------------------------------
    for (Op &o : t->ops) {
        Tensor &in = t->tensors[o.in0];
        Tensor &out = t->tensors[o.out];
        const size_t nin = (size_t)B * in.H * in.W * in.C, nout = (size_t)B * out.H * out.W * out.C;
        switch (o.kind) {
        case OP_CONV: {
            const size_t M = (size_t)B * o.H * o.W;
            const int Kc = o.kh * o.kw * o.Cin;
            if (fast_now(o)) {
                const float *am = in.am;
                if (!am) {
                    float *scratch = t->amax + (size_t)(2 * o.job) * B;
                    rc = amt_convh_absmax(in.v, nin / B, B, scratch, st);
                    if (rc != AMT_OK) return rc;
                    am = scratch;
                }
                rc = amt_convh_run(&o.fplan, in.v, out.v, nullptr, B, o.wp_f, t->sw + o.job, t->params[o.p1].w, am,
                                   (o.kh - 1) / 2, (o.kw - 1) / 2, st);
                if (rc != AMT_OK) return rc;
                break;
            }
            const float *A = in.v;
            if (!(o.kh == 1 && o.kw == 1)) {
                im2col_kernel<<<(unsigned)std::min<size_t>(M, 1u << 20), 256, 0, st>>>(in.v, (size_t)o.H * o.W * o.Cin, B, o.H, o.W, o.Cin, o.kh, o.kw, t->col);
                A = t->col;
            }
            rc = gemm(t, false, false, A, Kc, t->params[o.p0].w, o.Cout, out.v, o.Cout, (int)M, o.Cout, Kc, t->params[o.p1].w, st);
            if (rc != AMT_OK) return rc;
            break;
        }
        case OP_BN: {
            const size_t M = nin / o.Cout;
            if (o.bn_fused) {
                const int nwg = bnf_nwg(M, o.Cout);
                if (training) {
                    bnf_stats_kernel<<<nwg, 256, 0, st>>>(in.v, M, o.Cout, t->red0, t->red1);
                    bnf_stats_final_kernel<<<(o.Cout + 31) / 32, 256, 0, st>>>(
                        t->red0, t->red1, M, o.Cout, (int)bnf_rows_wg(o.Cout), nwg, o.bmu, o.bvar, o.binv, t->params[o.p2].w,
                        t->params[o.p3].w, 1, M > 1 ? (float)((double)M / ((double)M - 1.0)) : 1.0f);
                } else {
                    AMT_HIP_CHECK(hipMemcpyAsync(o.bmu, t->params[o.p2].w, o.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
                    inv_from_var_kernel<<<(o.Cout + 63) / 64, 64, 0, st>>>(t->params[o.p3].w, o.Cout, o.binv);
                }
                Tensor &dst = t->tensors[o.sig_out >= 0 ? o.sig_out : o.out];
                bnf_act_kernel<<<nwg, 256, 0, st>>>(in.v, o.bmu, o.binv, t->params[o.p0].w, t->params[o.p1].w, M, o.Cout,
                                                    in.H * in.W, o.sig_out >= 0 ? 1 : 0, dst.v, dst.am_slot);
                dst.am = dst.am_slot;
                break;
            }
            if (training) {
                rc = colreduce(t, in.v, nullptr, nullptr, nullptr, M, o.Cout, 0, o.bmu, nullptr, st);
                if (rc == AMT_OK) rc = colreduce(t, in.v, nullptr, o.bmu, nullptr, M, o.Cout, 1, o.bvar, o.binv, st);
                if (rc != AMT_OK) return rc;
                moving_update_kernel<<<(o.Cout + 63) / 64, 64, 0, st>>>(t->params[o.p2].w, t->params[o.p3].w, o.bmu, o.bvar, o.Cout,
                                                                       M > 1 ? (float)((double)M / ((double)M - 1.0)) : 1.0f);
            } else {
                AMT_HIP_CHECK(hipMemcpyAsync(o.bmu, t->params[o.p2].w, o.Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
                inv_from_var_kernel<<<(o.Cout + 63) / 64, 64, 0, st>>>(t->params[o.p3].w, o.Cout, o.binv);
            }
            bn_apply_kernel<<<grid1(nin), 256, 0, st>>>(in.v, o.bmu, o.binv, t->params[o.p0].w, t->params[o.p1].w, nin, o.Cout, out.v);
            break;
        }
        case OP_SIGMOID:
            if (o.skip) break;
            sigmoid_fwd_kernel<<<grid1(nin), 256, 0, st>>>(in.v, nin, out.v);
            break;
        case OP_ADD:
            add_kernel<<<grid1(nin), 256, 0, st>>>(in.v, t->tensors[o.in1].v, nin, out.v);
            break;
        case OP_POOL:
            pool_fwd_kernel<<<grid1(nout), 256, 0, st>>>(in.v, B, o.H, o.W, o.Cin, o.kh, o.kw, o.is_max, out.v,
                                                          (size_t)out.H * out.W * out.C);
            out.am = in.am;                         // a max / average of values bounded by the window's maximum stays bounded by it
            break;
        case OP_FLATTEN:
            copy_rows_kernel<<<grid1(nin), 256, 0, st>>>(in.v, (size_t)in.H * in.W * in.C, out.v + o.flat_off, (size_t)t->flat,
                                                          B, (size_t)in.H * in.W * in.C);
            break;
        case OP_DENSE:
            rc = gemm(t, false, false, in.v, o.Cin, t->params[o.p0].w, o.Cout, out.v, o.Cout, B, o.Cout, o.Cin, t->params[o.p1].w, st);
            if (rc != AMT_OK) return rc;
            break;
        }
        mark("fwd", o);
        ++op_no;
    }
    AMT_LAUNCH_CHECK();
    // ---------------- loss --------------------------------------------------------------------
    loss_kernel<<<(B + 63) / 64, 64, 0, st>>>(t->tensors[t->t_logits].v, y, B, K, t->pred, t->dlogits, t->loss_rows);
    if (pred_out) AMT_HIP_CHECK(hipMemcpyAsync(pred_out, t->pred, (size_t)B * K * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (loss_host) {
        std::vector<float> rows(B);
        AMT_HIP_CHECK(hipMemcpyAsync(rows.data(), t->loss_rows, B * sizeof(float), hipMemcpyDeviceToHost, st));
        AMT_HIP_CHECK(hipStreamSynchronize(st));
        double s = 0;
        for (float v : rows) s += v;
        *loss_host = (float)(s / B);
    }
    if (!update) return AMT_OK;
    // ---------------- backward ----------------------------------------------------------------
    rc = give_view(t, t->t_logits, t->dlogits, (size_t)B * K, st);
    if (rc != AMT_OK) return rc;
    for (int oi = (int)t->ops.size() - 1; oi >= 0; --oi) {
        Op &o = t->ops[oi];
        Tensor &in = t->tensors[o.in0];
        Tensor &out = t->tensors[o.out];
        const size_t nin = (size_t)B * in.H * in.W * in.C;
        if (o.kind == OP_SIGMOID && o.skip) continue;                 // folded into the BatchNormalization in front
        const int gsrc = (o.kind == OP_BN && o.sig_out >= 0) ? o.sig_out : o.out;   // tensor whose gradient arrives here
        if (o.kind != OP_FLATTEN && !t->tensors[gsrc].g_set) continue;               // no gradient reaches this op
        const float *gout = t->tensors[gsrc].gptr;
        op_no = oi;
        mark("bwd-enter", o);
        switch (o.kind) {
        case OP_DENSE: {
            // dW = in^T dOut, db = colsum dOut, dIn = dOut W^T
            rc = gemm(t, true, false, in.v, o.Cin, gout, o.Cout, t->params[o.p0].g, o.Cout, o.Cin, o.Cout, B, nullptr, st);
            if (rc == AMT_OK) rc = colreduce(t, gout, nullptr, nullptr, nullptr, (size_t)B, o.Cout, 3, t->params[o.p1].g, nullptr, st);
            if (rc != AMT_OK) return rc;
            if ((size_t)B * o.Cin > t->col_cap) return AMT_E_NOMEM;
            rc = gemm(t, false, true, gout, o.Cout, t->params[o.p0].w, o.Cout, t->col, o.Cin, B, o.Cin, o.Cout, nullptr, st);
            if (rc == AMT_OK) rc = give_copy(t, o.in0, t->col, (size_t)B * o.Cin, st);
            if (rc != AMT_OK) return rc;
            break;
        }
        case OP_FLATTEN: {
            Tensor &fl = t->tensors[o.out];
            if (!fl.g_set) break;
            // rows of the dense input's gradient -> this tower's last activation (written straight into its grad)
            copy_rows_kernel<<<grid1(nin), 256, 0, st>>>(fl.gptr + o.flat_off, (size_t)t->flat, in.g, (size_t)in.H * in.W * in.C, B,
                                                          (size_t)in.H * in.W * in.C);
            in.gptr = in.g; in.g_set = true;
            break;
        }
        case OP_SIGMOID: {
            const float *acc = nullptr;
            float *dst = claim(t, o.in0, &acc);
            if (acc) {
                if (nin > t->col_cap) return AMT_E_NOMEM;
                sigmoid_bwd_kernel<<<grid1(nin), 256, 0, st>>>(gout, out.v, nin, t->col);
                add_kernel<<<grid1(nin), 256, 0, st>>>(acc, t->col, nin, dst);
            } else {
                sigmoid_bwd_kernel<<<grid1(nin), 256, 0, st>>>(gout, out.v, nin, dst);
            }
            break;
        }
        case OP_ADD: {
            rc = give_view(t, o.in0, gout, nin, st);
            if (rc == AMT_OK) rc = give_view(t, o.in1, gout, nin, st);
            if (rc != AMT_OK) return rc;
            break;
        }
        case OP_POOL: {
            const float *acc = nullptr;
            float *dst = claim(t, o.in0, &acc);
            if (acc) {
                if (nin > t->col_cap) return AMT_E_NOMEM;
                pool_bwd_kernel<<<grid1(nin), 256, 0, st>>>(gout, (size_t)out.H * out.W * out.C, in.v, B, o.H, o.W, o.Cin, o.kh, o.kw,
                                                             o.is_max, t->col);
                add_kernel<<<grid1(nin), 256, 0, st>>>(acc, t->col, nin, dst);
            } else {
                pool_bwd_kernel<<<grid1(nin), 256, 0, st>>>(gout, (size_t)out.H * out.W * out.C, in.v, B, o.H, o.W, o.Cin, o.kh, o.kw,
                                                             o.is_max, dst);
            }
            break;
        }
        case OP_BN: {
            const size_t M = nin / o.Cout;
            if (o.bn_fused) {
                // dgamma = sum dy zhat, dbeta = sum dy, with dy = dA a (1 - a) when the sigmoid is folded in
                const int nwg = bnf_nwg(M, o.Cout);
                const float *a = o.sig_out >= 0 ? t->tensors[o.sig_out].v : nullptr;
                bnf_bwd_reduce_kernel<<<nwg, 256, 0, st>>>(gout, a, in.v, o.bmu, o.binv, M, o.Cout, t->red0, t->red1);
                colreduce_final_kernel<<<(o.Cout + 31) / 32, 256, 0, st>>>(t->red0, t->red1, o.Cout, 1.0f / (float)M, 2, nwg,
                                                                             t->params[o.p1].g, t->params[o.p0].g);
                const float *acc = nullptr;
                float *dst = claim(t, o.in0, &acc);
                bnf_bwd_apply_kernel<<<nwg, 256, 0, st>>>(gout, a, in.v, o.bmu, o.binv, t->params[o.p0].w, t->params[o.p1].g,
                                                          t->params[o.p0].g, 1.0f / (float)M, M, o.Cout, in.H * in.W, acc, dst,
                                                          t->red2, in.gam_slot);
                // red2 now holds THIS launch's column partials: valid as the bias gradient of the convolution in front only when
                // dz was written, not accumulated; any partials another convolution was still waiting for are gone either way
                if (!acc) { in.gam = in.gam_slot; t->colsum_of = o.in0; t->colsum_nwg = nwg; }
                else t->colsum_of = -1;
                break;
            }
            // generic path: dy -> scratch, two column sums, dz
            if (nin > t->col_cap) return AMT_E_NOMEM;
            AMT_HIP_CHECK(hipMemcpyAsync(t->col, gout, nin * sizeof(float), hipMemcpyDeviceToDevice, st));
            rc = colreduce(t, t->col, in.v, o.bmu, o.binv, M, o.Cout, 2, t->params[o.p1].g, t->params[o.p0].g, st);
            if (rc != AMT_OK) return rc;
            bn_backward_kernel<<<grid1(nin), 256, 0, st>>>(t->col, in.v, o.bmu, o.binv, t->params[o.p0].w, t->params[o.p1].g,
                                                            t->params[o.p0].g, 1.0f / (float)M, nin, o.Cout, 1, t->col);
            rc = give_copy(t, o.in0, t->col, nin, st);
            if (rc != AMT_OK) return rc;
            break;
        }
        case OP_CONV: {
            const size_t M = (size_t)B * o.H * o.W;
            const int Kc = o.kh * o.kw * o.Cin;
            const bool one = o.kh == 1 && o.kw == 1;
            if (t->wgrad_direct_on && wgrad_direct_ok(o) && M >= t->fast_min_m) {
                rc = wgrad_direct(t, o, in.v, gout, t->params[o.p0].g, B, st);
            } else {
                const float *A = in.v;
                if (!one) {
                    im2col_kernel<<<(unsigned)std::min<size_t>(M, 1u << 20), 256, 0, st>>>(in.v, (size_t)o.H * o.W * o.Cin, B, o.H, o.W, o.Cin, o.kh, o.kw, t->col);
                    A = t->col;
                }
                rc = gemm(t, true, false, A, Kc, gout, o.Cout, t->params[o.p0].g, o.Cout, Kc, o.Cout, (int)M, nullptr, st);
            }
            if (rc != AMT_OK) return rc;
            // bias gradient: the column sums of dz -- per-workgroup partials are there when the BatchNormalization behind
            // this convolution just wrote dz
            if (t->colsum_of == o.out) {
                colreduce_final_kernel<<<(o.Cout + 31) / 32, 256, 0, st>>>(t->red2, t->red2, o.Cout, 1.0f / (float)M, 3, t->colsum_nwg,
                                                                             t->params[o.p1].g, nullptr);
                t->colsum_of = -1;
            } else {
                rc = colreduce(t, gout, nullptr, nullptr, nullptr, M, o.Cout, 3, t->params[o.p1].g, nullptr, st);
                if (rc != AMT_OK) return rc;
            }
            const bool is_input = std::find(t->inputs.begin(), t->inputs.end(), o.in0) != t->inputs.end();
            if (is_input) break;                                        // no gradient wrt the network input
            if (o.fast_bwd && fast_now(o)) {
                // dX = correlation of dOut with the flipped, transposed kernel (mirror-image padding), written into the
                // input's gradient buffer, plus what the shortcut branch already handed to that tensor
                const float *am = out.gam;
                if (!am) {
                    float *scratch = t->amax + (size_t)(2 * o.job + 1) * B;
                    rc = amt_convh_absmax(gout, (size_t)o.H * o.W * o.Cout, B, scratch, st);
                    if (rc != AMT_OK) return rc;
                    am = scratch;
                }
                const float *acc = nullptr;
                float *dst = claim(t, o.in0, &acc);
                rc = amt_convh_run(&o.bplan, gout, dst, acc, B, o.wp_b, t->sw + o.job, nullptr, am,
                                   o.kh - 1 - (o.kh - 1) / 2, o.kw - 1 - (o.kw - 1) / 2, st);
                if (rc != AMT_OK) return rc;
                break;
            }
            // dcol = dOut W^T (into col), then gathered back to the image
            rc = gemm(t, false, true, gout, o.Cout, t->params[o.p0].w, o.Cout, t->col, Kc, (int)M, Kc, o.Cout, nullptr, st);
            if (rc != AMT_OK) return rc;
            if (one) rc = give_copy(t, o.in0, t->col, nin, st);
            else {
                Tensor &inT = t->tensors[o.in0];
                if (!inT.g_set) {
                    col2im_kernel<<<(unsigned)std::min<size_t>(M, 1u << 20), 256, 0, st>>>(t->col, B, o.H, o.W, o.Cin, o.kh, o.kw, inT.g);
                    inT.gptr = inT.g; inT.g_set = true; inT.gam = nullptr;
                } else {
                    // dOut's own buffer is dead once both GEMMs above have read it: gather there, then add
                    if (gout != out.g || (size_t)B * out.H * out.W * out.C < nin) return AMT_E_NOMEM;
                    col2im_kernel<<<(unsigned)std::min<size_t>(M, 1u << 20), 256, 0, st>>>(t->col, B, o.H, o.W, o.Cin, o.kh, o.kw, out.g);
                    rc = give_copy(t, o.in0, out.g, nin, st);
                }
            }
            if (rc != AMT_OK) return rc;
            break;
        }
        }
    }
    AMT_LAUNCH_CHECK();
    // ---------------- update -------------------------------------------------------------------
    adagrad_kernel<<<grid1(t->arena_used), 256, 0, st>>>(t->w_arena, t->g_arena, t->a_arena, t->arena_used, t->lr, t->eps);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

static int copy_out(amt_trainer *t, float *host, size_t n, bool grads) {
    if (!t || !host) return AMT_E_INVALID;
    size_t total = 0;
    for (const Param &p : t->params) total += p.n;
    if (total != n) return AMT_E_SHAPE;
    AMT_HIP_CHECK(hipDeviceSynchronize());
    float *cur = host;
    for (const Param &p : t->params) {
        if (grads && !p.trainable) memset(cur, 0, p.n * sizeof(float));
        else AMT_HIP_CHECK(hipMemcpy(cur, grads ? p.g : p.w, p.n * sizeof(float), hipMemcpyDeviceToHost));
        cur += p.n;
    }
    return AMT_OK;
}
int amt_trainer_get_weights(amt_trainer *t, float *weights_host, size_t n_floats) { return copy_out(t, weights_host, n_floats, false); }
int amt_trainer_get_grads(amt_trainer *t, float *grads_host, size_t n_floats) { return copy_out(t, grads_host, n_floats, true); }

}  // extern "C"
