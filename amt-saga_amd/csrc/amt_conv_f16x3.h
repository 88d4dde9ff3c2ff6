// Split-fp16 convolution ("f16x3"): fp32-equivalent products on the f16 MFMA pipe with
// THREE f16 MFMAs per 16-deep k-block of an f32 product (the split-bf16 kernel needs six).
//
// Every f32 operand x (scaled by a power of two into the f16 range) is written as
//        x * 2^s = h + l * 2^-11,   h = f16(x 2^s),  l = f16((x 2^s - h) * 2^11)
// h carries 11 significand bits and l the next 11 (x 2^s - h is exact in f32, Sterbenz), so
// the pair represents x to <= 2^-23 relative -- the f32 rounding unit.  A product is
//        a b = ha hb + 2^-11 (ha lb + la hb) + 2^-22 la lb
// the first three terms are exact f16 x f16 products accumulated in f32 by the MFMA (two
// accumulators: `hi` for ha hb, `lo` for the two cross terms, combined once at the end);
// the dropped la lb term is <= 2^-22 |ab|.  Per product: <= 2^-21 relative in the worst case
// (two representation errors + the dropped term), ~1e-7 rms -- a few f32 ulps; a K = 2048
// contraction lands as close to the float64 result as a plain f32 dot product does
// (tests/test_split_fp16_bound.py), far inside the 1e-4 parity bound and the 5e-6 the tests hold
// the heads' logits to.
//
// Range.  f16 spans 2^-14 .. 65504, so operands are pre-scaled by exact powers of two:
//   * weights: per layer, 2^sw with max|w| 2^sw in [8, 16)                         (host)
//   * activations: per layer, per launch and PER WINDOW, 2^sa with  amax * 2^sa < 2^13, where amax is
//     the MEASURED max |input| of that window: every producer of a convolution input (absmax_kernel for
//     the network input, the epilogues of conv1_mfma_kernel and of this kernel) leaves max |output| per
//     window in a device array (atomicMax of non-negative float bits), which the consumer reads.  A
//     max-pool between two layers keeps the bound valid.  Measured, not derived from the folded BN
//     parameters: a static bound compounds over the residual blocks and, with realistic (trained-like) BN
//     statistics, ends up orders of magnitude above the activations, wasting the f16 range; and per
//     window, so that a window's result does not depend on the other windows of its batch.
//   2^-(sa+sw) is applied per output row in the epilogue (exact).  Scaled values below
//   2^-14 (f16 subnormals) are flushed to zero in both terms: an absolute error below
//   2^-25 of the scaled bound, i.e. < 2^-38 relative to the window's largest activation.
//
// conv_f16x3s_kernel (512 threads, two workgroups per CU, <= 128 VGPRs, spill-free): input tile
// [pos][plane(2)][16 ch] f16 with an 80-B pitch = 5 x 16-B slots, split while staging; weights
// global -> registers -> double-buffered LDS in 16-KB groups, one barrier per group; a wave owns ONE
// 32-position M-tile and 32 output channels (wider layers are split over blockIdx.y) and walks all
// taps with v_mfma_f32_16x16x32_f16 on tap pairs; the next chunk's input is requested under the last
// weight group; small images (H x W <= 64) use the masked form (no halo).  Tile geometry and the
// masked idea follow conv_bf16x6_kernel (amt_rdcnn.hip).  An earlier two-M-tiles-per-wave kernel on
// v_mfma_f32_32x32x16_f16 and the experiments around it are recorded in profiles/r01/ablation_f16x3.txt.
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HX_PSTRIDE 80                        // bytes per staged position: 2 planes x 16 ch x 2 B + 16 pad
#define HXS_NIT 2                            // prefetched staging items per thread (conv_f16x3s_kernel); 3 measured slower (register pressure)
#define HX_TPITCH 36                         // floats per row of the epilogue's transposition patch
#define HX_LSCALE 2048.0f                    // 2^11
#define HX_MINNORM 6.103515625e-05f          // 2^-14

__host__ __device__ __forceinline__ unsigned short amt_f16_bits(_Float16 h) {
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
// x is already scaled.  Returns the two f16 bit patterns.  FLUSH = true (host, weights): a term
// below the f16 normal range is set to zero, so the result does not depend on how the MFMA treats
// f16 subnormals (for h < 2^-14 the whole value then moves into l, which has 2^11 more range).
// FLUSH = false (device, activations: 4 VALU ops fewer per staged element): if the matrix pipe
// flushed a subnormal term the value error would be < 2^-14 in scaled units = 2^-27 of the
// layer's activation bound -- not worth the instructions.
template <bool FLUSH>
__host__ __device__ __forceinline__ void amt_split_f16(float xs, unsigned short &hb, unsigned short &lb) {
    _Float16 h = (_Float16)xs;
    if (FLUSH && !(__builtin_fabsf(xs) >= HX_MINNORM)) h = (_Float16)0.0f;
    const float r = (xs - (float)h) * HX_LSCALE;
    _Float16 l = (_Float16)r;
    if (FLUSH && !(__builtin_fabsf(r) >= HX_MINNORM)) l = (_Float16)0.0f;
    hb = amt_f16_bits(h);
    lb = amt_f16_bits(l);
}

struct HxScale {
    const float *amax_in;     // device [B]: max |input| of every window of this launch
    float *amax_out;          // device [B] or null: max |output| per window (atomicMax, zeroed by the caller)
    int sw;                   // weights were scaled by 2^sw on the host
    unsigned long long *ts;   // diagnostic (AMT_CONV_TS): four 100-MHz timestamps per workgroup, or null
    const int *sw_dev = nullptr;   // TRAIN form: the weight exponent lives on the device (the weights move every step)
};
#define HX_MAXWIN 64                         // windows per workgroup tile (masked tiles of >= 4 positions)

// exponent sa with amax * 2^sa < 2^13 (0 for an all-zero or non-finite window; clamped so that
// 2^-(sa+sw) stays a normal float)
__device__ __forceinline__ int hx_scale_exp(float amax) {
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &e);
    else return 0;
    return min(max(13 - e, -90), 90);
}

// floor(q / d) for 0 <= q < 2^20 with rd = 1.0f / d: (q + 0.5) rd is off by < 2^-3 / d from (q + 0.5) / d, which
// lies at least 0.5 / d from the next integer.  Two instructions instead of the ~40 of an integer division: the
// tile tables sit in front of the first load of a workgroup.
__device__ __forceinline__ int hx_div(int q, float rd) { return (int)(((float)q + 0.5f) * rd); }

// max |x| per window: grid (blocks, B); out[b] must be zero before
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, size_t stride,
                                                      float *__restrict__ out) {
    __shared__ float red[16];
    const float *xb = x + (size_t)blockIdx.y * stride;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(xb[i]));                      // fmaxf drops NaNs
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int *>(out) + blockIdx.y, __float_as_int(m));
}

// ---------------------------------------------------------------------------------------------
// One 32-position M-tile and 32 output channels per wave, every tap.
//
// In conv_f16x3_kernel a wave holds four accumulators (two M-tiles x hi/lo = 64 VGPRs), which
// leaves hipcc one fragment register set: it reads an A fragment, waits, issues one or two
// MFMAs, reads the next -- an LDS round trip in front of nearly every MFMA pair.  With a single
// 32-channel N-tile the K-split that kept eight waves busy on 256 positions is not needed either:
// here wave w simply owns positions 32w .. 32w+31 and walks all taps (accumulators: 32 VGPRs), and
// the freed registers hold TWO taps of fragments (A h/l + B h/l = 16 VGPRs each): while the three
// MFMAs of tap t run, the four ds_read_b128 of tap t+1 are already in flight
// (sched_barriers keep hipcc from sinking them back).  No partial-sum exchange, same tile
// geometry, staging, weight groups (8 taps = 16 KB) and epilogue.  A B fragment now feeds one
// tile instead of two: 4 reads per 3 MFMAs, still well inside the LDS rate.
// ---------------------------------------------------------------------------------------------
// MASKED = true is the small-image form: whole H x W <= 64 images of several windows per workgroup,
// no halo in LDS; every tap's A-fragment address is chosen per lane (in-bounds neighbour or one
// all-zero position), so the "same" padding costs no LDS.
// MS = 16-position M-subtiles per wave: 2 (eight waves, 512 threads, four waves per SIMD) or 4 (four waves of 64
// positions, 256 threads, two waves per SIMD with twice the registers: a B fragment then feeds four M-subtiles
// instead of two -- 12 ds_read_b128 per 24 MFMAs instead of 8 per 12).
// TRAIN = true is the trainer's form (amt_train.hip: forward convolution and data gradient of a training step): the
// epilogue leaves the raw sum (+ bias, + an optional accumulate-into tensor through the shortcut slot) instead of
// sigmoid(BN(.)), the padding before the image is a runtime value (the data gradient is the correlation with the
// flipped kernel, whose "same" padding is the mirror image: KH - 1 - (KH - 1) / 2 rows before), and the weight
// exponent is read from the device.  The inference instantiations (TRAIN = false) compile exactly as before.
template <int KH, int KW, int CIN, bool MASKED, int MS = 2, bool TRAIN = false>
__global__ __launch_bounds__(1024 / MS, MS == 2 ? 4 : 2) void conv_f16x3s_kernel(ConvParams p, const uint4 *__restrict__ w16s,
                                                                                  HxScale hs) {
    constexpr int NT = 1024 / MS;                                   // threads per workgroup (256 positions / (16 MS) waves)
    constexpr int NIT = MS == 2 ? HXS_NIT : 4;                      // prefetched staging items per thread
    constexpr int NCHUNK = CIN / BX_CC;
    constexpr int NTAPS = KH * KW;
    constexpr int PCAP = 256;
    const int PAD_T = TRAIN ? p.pad_t : (KH - 1) / 2, PAD_L = TRAIN ? p.pad_l : (KW - 1) / 2;
    static_assert(NTAPS % 2 == 0, "tap pairs");
    constexpr int NSLAB = NTAPS / 2;                                // 4-KB slabs of two taps (host layout)
    constexpr int SLAB_V4 = 256;
    constexpr int GROUP = NSLAB >= 4 ? 4 : NSLAB;                   // slabs per weight group
    constexpr int GT = 2 * GROUP;                                   // taps per group
    constexpr int NG = NSLAB / GROUP, NGT = NCHUNK * NG;
    constexpr int GV4 = GROUP * SLAB_V4;
    constexpr int WPT = GV4 / NT;
    static_assert(NSLAB % GROUP == 0 && GV4 % NT == 0, "whole groups");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4 *wbuf = reinterpret_cast<uint4 *>(smem);                  // [2][GV4]
    int *pos_sp = reinterpret_cast<int *>(wbuf + 2 * GV4);          // [PCAP]
    int *pos_win = pos_sp + PCAP;
    float *pos_os = reinterpret_cast<float *>(pos_win + PCAP);      // [PCAP] 2^-(sa + sw) of the row's window
    float *win_is = pos_os + PCAP;                                  // [HX_MAXWIN] 2^sa of the tile's windows
    int *win_max = reinterpret_cast<int *>(win_is + HX_MAXWIN);     // [HX_MAXWIN] max |output| (float bits)
    float *bnp = reinterpret_cast<float *>(win_max + HX_MAXWIN);    // [4][32] s1, t1, s2, t2 of this 32-channel slice
    char *in_lds = reinterpret_cast<char *>(bnp + 128);             // [POSIN][80 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long *tsp = hs.ts ? hs.ts + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (tsp && tid == 0) tsp[0] = wall_clock64();
    const int THin = MASKED ? p.TH : p.TH + KH - 1, TWin = MASKED ? p.TW : p.TW + KW - 1;
    const int RP = MASKED ? p.TW : bx_row_pitch(p.TW, TWin);
    const int cout_off = blockIdx.y * 32;
    const uint4 *w16 = w16s + (size_t)blockIdx.y * ((size_t)NCHUNK * NSLAB * SLAB_V4);
    int bid;
    {
        const int nx = gridDim.x, q8 = nx >> 3, r8 = nx & 7, xcd = blockIdx.x & 7;
        bid = xcd * q8 + min(xcd, r8) + (blockIdx.x >> 3);
    }
    const int tc = bid % p.tiles_w; bid /= p.tiles_w;
    const int tr = bid % p.tiles_h; bid /= p.tiles_h;
    const int win0 = bid * p.NWIN;
    const int r0 = tr * p.TH, c0 = tc * p.TW;
    const int ptile = p.TH * p.TW;
    const float rd_ptile = 1.0f / (float)ptile, rd_tw = 1.0f / (float)p.TW;
    const float rd_twin = 1.0f / (float)TWin, rd_thin = 1.0f / (float)THin;

    // per-window scales of this tile and the folded BN parameters of this channel slice: requested first, so that
    // their latency runs with the weight / tile loads issued below; the tables are written after those
    float my_amax = 0.f;
    if (tid < p.NWIN && win0 + tid < p.B) my_amax = hs.amax_in[win0 + tid];
    float my_bn = 0.f;
    if (tid < 128) {
        const float *src = tid < 32 ? p.s1 : tid < 64 ? p.t1 : tid < 96 ? p.s2 : p.t2;
        my_bn = src ? src[cout_off + (tid & 31)] : ((tid < 32 || (tid >= 64 && tid < 96)) ? 1.f : 0.f);   // absent: scale 1, shift 0
    }
    // v_mfma_f32_16x16x32_f16: A lane l = row l%16, k-group l/16 (8 k each); one MFMA contracts a
    // PAIR of taps x 16 channels: k-group g = (tap t + g%2, channels 8*(g/2) .. +7); the second tap
    // is the next column (+one position pitch).  With this order the lanes a ds_read_b128 services
    // together (0-3,12-15,20-27 ...) read 16 distinct slots or the same address: conflict-free.  Two 16-position M-subtiles x two 16-channel
    // N-subtiles x (hi, lo) = eight 4-register accumulators.
    int abase[MS];
    unsigned lrc[MS];                                      // MASKED: (row, col) of the M-subtiles, 8 bits each
    const int tsel = (lane >> 4) & 1;                      // this lane's tap of a tap pair
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        int q = wid * (16 * MS) + ms * 16 + (lane & 15);
        int w_ = hx_div(q, rd_ptile), rem = q - w_ * ptile;
        int r = hx_div(rem, rd_tw), c = rem - r * p.TW;
        if (w_ >= p.NWIN) { w_ = 0; r = 0; c = 0; }
        abase[ms] = ((w_ * THin + r) * RP + c) * HX_PSTRIDE + (lane >> 5) * 16 + (MASKED ? 0 : tsel * HX_PSTRIDE);
        lrc[ms] = (unsigned)r | ((unsigned)c << 8);
    }
    const int zero_off = p.NWIN * THin * RP * HX_PSTRIDE + (lane >> 5) * 16;
    if (MASKED && tid < 20) reinterpret_cast<unsigned *>(in_lds + p.NWIN * THin * RP * HX_PSTRIDE)[tid] = 0u;
    f32x4 hi[MS][2], lo[MS][2];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int e = 0; e < 4; ++e) { hi[ms][ns][e] = 0.f; lo[ms][ns][e] = 0.f; }

    u32x4 wp[WPT];
    auto issue = [&](int gg) {
        const uint4 *src = w16 + (size_t)gg * GV4 + tid;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wp[i]) : "v"(src + i * NT) : "memory");
    };
    // staged 32-byte items of this thread (the same for every chunk): up to HXS_NIT of them are
    // prefetched; the third slot exists only in the waves that have one (tiles of 1025..1536 items)
    const int items = p.NWIN * THin * TWin * 2;
    int sdst[NIT];                                         // LDS offset, -1: no item
    const float *ssrc[NIT];                                // chunk-0 source, null: outside the image (zeros)
    auto slot_live = [&](int u) { return u < 2 || u * NT + wid * 64 < items; };     // wave-uniform: slots past the tile
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int it = u * NT + tid;
        sdst[u] = -1; ssrc[u] = nullptr;
        if (it < items) {
            const int cg = it & 1, pc = it >> 1;
            const int wr = hx_div(pc, rd_twin), ci = pc - wr * TWin;
            const int w_ = hx_div(wr, rd_thin), ri = wr - w_ * THin;
            const int gr = r0 + ri - (MASKED ? 0 : PAD_T), gc = c0 + ci - (MASKED ? 0 : PAD_L), gw = win0 + w_;
            sdst[u] = ((wr * RP + ci) * HX_PSTRIDE + cg * 16) | (w_ << 24);       // LDS offset | window slot
            if (gr >= 0 && gr < p.H && gw < p.B && gc >= 0 && gc < p.W)
                ssrc[u] = p.in + (size_t)gw * p.in_win_stride + ((size_t)gr * p.W + gc) * CIN + cg * 8;
        }
    }
    u32x4 sv[NIT][2];                                      // staged values in flight (inline-asm loads)
    auto stage_issue = [&](int ch) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            if (!slot_live(u)) continue;
            const float *src = ssrc[u] ? ssrc[u] + ch * BX_CC : p.in;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sv[u][0]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(sv[u][1]) : "v"(src) : "memory");
        }
    };
    auto split_store = [&](const float (&v)[8], int dsto, float in_scale) {
        unsigned short h[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) amt_split_f16<false>(v[e] * in_scale, h[0][e], h[1][e]);
        char *dst = in_lds + dsto;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint4 pk;
            pk.x = h[pl][0] | ((unsigned)h[pl][1] << 16);
            pk.y = h[pl][2] | ((unsigned)h[pl][3] << 16);
            pk.z = h[pl][4] | ((unsigned)h[pl][5] << 16);
            pk.w = h[pl][6] | ((unsigned)h[pl][7] << 16);
            *reinterpret_cast<uint4 *>(dst + pl * 32) = pk;
        }
    };
    issue(0);                                             // first weight group: lands while the first tile is staged
    if constexpr (!MASKED) stage_issue(0);                 // (masked tiles are small: staged synchronously, registers saved)
    // everything below runs under the latency of the loads just issued
    for (int q = tid; q < PCAP; q += NT) {
        const int w_ = hx_div(q, rd_ptile), rem = q - w_ * ptile;
        const int r = hx_div(rem, rd_tw), c = rem - r * p.TW;
        const bool ok = w_ < p.NWIN && (win0 + w_) < p.B && (r0 + r) < p.H && (c0 + c) < p.W;
        pos_sp[q] = ok ? (r0 + r) * p.W + (c0 + c) : -1;
        pos_win[q] = win0 + w_;
    }
    if (tid < HX_MAXWIN) {
        win_is[tid] = __uint_as_float((unsigned)(127 + hx_scale_exp(my_amax)) << 23);     // 2^sa
        win_max[tid] = 0;
    }
    if (tid < 128) bnp[tid] = my_bn;
    for (int ch = 0; ch < NCHUNK; ++ch) {
        __syncthreads();
        // ---- stage + split the input tile (16 channels).  The first pass (two items per thread) was
        // issued ahead: for chunk 0 at kernel entry, for later chunks before the last weight group of
        // the previous chunk, so its latency runs under that group's MFMAs. ------------------------
        // chunk 0: wait here; later chunks: the values landed with the wait that closed the previous
        // chunk's last weight group (waiting again would expose the weight loads issued since)
        if constexpr (!MASKED) {
        if (ch == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < NIT; ++u) asm volatile("" : "+v"(sv[u][0]), "+v"(sv[u][1]));   // tie the values to the wait
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            if (sdst[u] < 0 || !slot_live(u)) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ssrc[u] ? __uint_as_float(sv[u][e >> 2][e & 3]) : 0.f;
            split_store(v, sdst[u] & 0xFFFFFF, win_is[sdst[u] >> 24]);
        }
        }
        for (int it = (MASKED ? 0 : NIT * NT) + tid; it < items; it += NT) { // masked tiles; items beyond the prefetch slots
            const int cg = it & 1, pc = it >> 1;
            const int wr = hx_div(pc, rd_twin), ci = pc - wr * TWin;
            const int w_ = hx_div(wr, rd_thin), ri = wr - w_ * THin;
            const int gr = r0 + ri - (MASKED ? 0 : PAD_T), gc = c0 + ci - (MASKED ? 0 : PAD_L), gw = win0 + w_;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (gr >= 0 && gr < p.H && gw < p.B && gc >= 0 && gc < p.W) {
                const float4 *src = reinterpret_cast<const float4 *>(
                    p.in + (size_t)gw * p.in_win_stride + ((size_t)gr * p.W + gc) * CIN + ch * BX_CC + cg * 8);
                const float4 x0 = src[0], x1 = src[1];
                v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            }
            split_store(v, (wr * RP + ci) * HX_PSTRIDE + cg * 16, win_is[w_]);
        }
        if (ch == 0) {                                    // weight group 0 -> LDS, group 1 -> prefetch registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < WPT; ++i) reinterpret_cast<u32x4 *>(wbuf)[tid + i * NT] = wp[i];
            if (NGT > 1) issue(1);
        }
        __syncthreads();                                   // tile staged, weight group parked
        if (tsp && tid == 0 && ch == 0) tsp[1] = wall_clock64();
        union U { uint4 u; f16x8 v; };
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int gg = ch * NG + g;
            const uint4 *wb = wbuf + (gg & 1) * GV4 + lane;          // [tap pair][plane][N-subtile][64 lanes]
            U a[2];                                                  // A fragments of one M-subtile [plane]
            U b[2][2];                                               // [N-subtile][plane]
            auto loadA = [&](int tp, int ms) {
                const int tap = g * GT + 2 * tp;
                const int dy = tap / KW, dx = tap - dy * KW;
                const char *ap;
                if constexpr (MASKED) {
                    const int rr = (int)(lrc[ms] & 255u) + dy - PAD_T;
                    const int cc = (int)((lrc[ms] >> 8) & 255u) + dx + tsel - PAD_L;
                    const bool inb = (unsigned)rr < (unsigned)p.H && (unsigned)cc < (unsigned)p.W;
                    ap = in_lds + (inb ? abase[ms] + ((dy - PAD_T) * RP + (dx + tsel - PAD_L)) * HX_PSTRIDE : zero_off);
                } else {
                    ap = in_lds + abase[ms] + (dy * RP + dx) * HX_PSTRIDE;
                }
                a[0].u = *reinterpret_cast<const uint4 *>(ap);
                a[1].u = *reinterpret_cast<const uint4 *>(ap + 32);
            };
            auto mfma6 = [&](int ms) {
#pragma unroll
                for (int ns = 0; ns < 2; ++ns) {
                    lo[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1].v, b[ns][0].v, lo[ms][ns], 0, 0, 0);
                    hi[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[ns][0].v, hi[ms][ns], 0, 0, 0);
                    lo[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0].v, b[ns][1].v, lo[ms][ns], 0, 0, 0);
                }
            };
            // next chunk's tile: loads fly under this (last) group's matrix work
            if constexpr (!MASKED) if (g == NG - 1 && ch + 1 < NCHUNK) stage_issue(ch + 1);
#pragma unroll
            for (int tp = 0; tp < GT / 2; ++tp) {
#pragma unroll
                for (int ns = 0; ns < 2; ++ns)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) b[ns][pl].u = wb[((tp * 2 + pl) * 2 + ns) * 64];
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    loadA(tp, ms);
                    mfma6(ms);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (gg + 1 < NGT) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // group gg+1 has landed in registers
                u32x4 *dst = reinterpret_cast<u32x4 *>(wbuf + ((gg + 1) & 1) * GV4);
#pragma unroll
                for (int i = 0; i < WPT; ++i) dst[tid + i * NT] = wp[i];
                if (gg + 2 < NGT) issue(gg + 2);
            }
            if (g + 1 < NG) __syncthreads();
        }
    }
    // ---- epilogue (see conv_f16x3_kernel) -------------------------------------------------------
    if (tsp && tid == 0) tsp[2] = wall_clock64();
    if (tid < PCAP) {                                      // 2^-(sa + sw) of every output row's window
        const int wl = min(max(pos_win[tid] - win0, 0), HX_MAXWIN - 1);
        const int sa = (int)(__float_as_uint(win_is[wl]) >> 23) - 127;
        const int sw = TRAIN ? *hs.sw_dev : hs.sw;
        pos_os[tid] = __uint_as_float((unsigned)(127 - (sa + sw)) << 23);
    }
    __syncthreads();
    float *tb = reinterpret_cast<float *>(in_lds) + wid * (16 * MS * HX_TPITCH);
    const int c4 = (lane & 7) * 4;
    const float4 s2v = *reinterpret_cast<const float4 *>(bnp + 64 + c4), t2v = *reinterpret_cast<const float4 *>(bnp + 96 + c4);
    // D of a 16x16 MFMA: lane l holds rows 4*(l/16) + e, column l%16
    float osc[MS][4];                                  // 2^-(sa + sw) of this lane's rows
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int e = 0; e < 4; ++e) osc[ms][e] = pos_os[wid * (16 * MS) + ms * 16 + 4 * (lane >> 4) + e];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
        const float s1 = bnp[ns * 16 + (lane & 15)], t1 = bnp[32 + ns * 16 + (lane & 15)];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ms * 16 + 4 * (lane >> 4) + e;
                const float z = (hi[ms][ns][e] + lo[ms][ns][e] * (1.0f / HX_LSCALE)) * osc[ms][e];
                tb[row * HX_TPITCH + ns * 16 + (lane & 15)] = TRAIN ? z + t1 : sigmoidf_(z * s1 + t1);
            }
    }
    constexpr int NR = 2 * MS;                             // eight rows of the wave's patch per pass
    int spq[NR], gwq[NR];
    float4 scv[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int q = wid * (16 * MS) + (lane >> 3) + 8 * i;
        spq[i] = pos_sp[q];
        gwq[i] = pos_win[q];
    }
    const bool has_sc = p.sc || p.sc1;
    if (p.sc) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float *scp = p.sc + (size_t)gwq[i] * p.sc_win_stride + (size_t)max(spq[i], 0) * p.cout_total + cout_off + c4;
            scv[i] = spq[i] >= 0 ? *reinterpret_cast<const float4 *>(scp) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if (p.sc1) {
        // rank-1 shortcut: BN(conv1x1(x)) of the one-channel input, formed here as proj_kernel forms it
        const float4 w4 = *reinterpret_cast<const float4 *>(p.sc1_w + cout_off + c4);
        const float4 s4 = *reinterpret_cast<const float4 *>(p.sc1_s + cout_off + c4);
        const float4 t4 = *reinterpret_cast<const float4 *>(p.sc1_t + cout_off + c4);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float x = spq[i] >= 0 ? p.sc1[(size_t)gwq[i] * p.sc1_win_stride + spq[i]] : 0.f;
            scv[i] = make_float4(fmaf(x, w4.x, 0.f) * s4.x + t4.x, fmaf(x, w4.y, 0.f) * s4.y + t4.y,
                                 fmaf(x, w4.z, 0.f) * s4.z + t4.z, fmaf(x, w4.w, 0.f) * s4.w + t4.w);
        }
    }
    float rmax[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float4 v = *reinterpret_cast<const float4 *>(tb + ((lane >> 3) + 8 * i) * HX_TPITCH + c4);
        rmax[i] = 0.f;
        if (spq[i] < 0) continue;
        if (has_sc) {
            v.x = (v.x + scv[i].x) * s2v.x + t2v.x;
            v.y = (v.y + scv[i].y) * s2v.y + t2v.y;
            v.z = (v.z + scv[i].z) * s2v.z + t2v.z;
            v.w = (v.w + scv[i].w) * s2v.w + t2v.w;
        }
        rmax[i] = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        *reinterpret_cast<float4 *>(p.out + (size_t)gwq[i] * p.out_win_stride + (size_t)spq[i] * p.cout_total + cout_off + c4) = v;
    }
    // max |output| per window for the next layer's operand scaling
    if (hs.amax_out) {
        if (p.NWIN == 1) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < NR; ++i) m = fmaxf(m, rmax[i]);
            m = wave_max(m);
            if (lane == 0) atomicMax(win_max, __float_as_int(m));
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                float m = rmax[i];                      // eight lanes share a row
                m = fmaxf(m, __shfl_xor(m, 1, 64));
                m = fmaxf(m, __shfl_xor(m, 2, 64));
                m = fmaxf(m, __shfl_xor(m, 4, 64));
                if ((lane & 7) == 0 && spq[i] >= 0) atomicMax(win_max + (gwq[i] - win0), __float_as_int(m));
            }
        }
        __syncthreads();
        if (tid < p.NWIN && win0 + tid < p.B) atomicMax(reinterpret_cast<int *>(hs.amax_out) + win0 + tid, win_max[tid]);
    }
    if (tsp && tid == 0) tsp[3] = wall_clock64();
}
