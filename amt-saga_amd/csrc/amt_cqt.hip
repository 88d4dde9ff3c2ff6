// Constant-Q slices at the (<= 8) frames the heads keep, for gfx950.
//
// Stands where audio_complete.slice_C calls |librosa.cqt| and keeps
// _resize(C[:, s:t], target) (/root/reference/util_audio.py:411-434).  The
// transform is the build's own spec (oracle/cqt.py): direct constant-Q
// response with periodic-Hann, L1-normalised filters of length N_k, scaled by
// sqrt(N_k); frequencies quantised to uint32 cycles/sample so the oscillator
// phase is exact integer arithmetic on CPU and GPU alike.
//
// One workgroup per (window, bin).  With w(n) = 1/2 - 1/2 cos(theta n), theta = 2 pi / N_k,
// a frame that starts o hops after the first one is
//     C_o = 1/2 S0 - 1/4 ( e^{-i theta o H} S+  +  e^{+i theta o H} S- )
//     S0 = sum x[m] e^{-i phi m},  S+- = sum x[m] e^{-i phi m} e^{+-i theta u},  u = m - a_min
// summed over the frame's support.  The frames of a slice start whole hops apart, so their
// supports are the same run of H-sample blocks shifted by o: the blocks every frame shares
// (the "core") are accumulated once in registers -- one oscillator sincos, one window sincos
// and a dozen FMAs per sample, independent of the number of frames -- and only the <= 7 head
// / <= 8 tail blocks (plus the N_k mod H prefix of the block behind each frame) are kept as
// per-block sums in LDS and combined per frame at the end.  Frame sets wider than 8 hops
// (the song-level normalisers sample the whole window) take the generic path: every frame's
// Hann weight per sample via the angle-sum identity.  VALU/L2-bound; no MFMA.
#include "amt_common.h"

#define AMT_CQT_MAXF 8
#define CQ_EDGE 16                      // head blocks 0..6 -> 0..6, tail blocks nb+i -> 7+i (i <= 8)

struct c3 { float r0, i0, rp, ip, rm, im; };

__device__ __forceinline__ void cq_accum(c3 &s, float xv, float c, float sn, float cw, float sw) {
    // z = x e^{-i phi} = (xv c, -xv sn).  S+- = sum z e^{+-i theta u} = P +- iQ with the cosine- and
    // sine-weighted sums P = sum z cos(theta u), Q = sum z sin(theta u): four multiply-adds per
    // sample instead of eight; (rp, ip) holds P and (rm, im) holds Q until cq_finish().
    const float zr = xv * c, zi = -xv * sn;
    s.r0 += zr; s.i0 += zi;
    s.rp += zr * cw; s.ip += zi * cw;
    s.rm += zr * sw; s.im += zi * sw;
}
__device__ __forceinline__ void cq_finish(c3 &s) {        // (P, Q) -> (S+, S-)
    const float pr = s.rp, pi_ = s.ip, qr = s.rm, qi = s.im;
    s.rp = pr - qi; s.ip = pi_ + qr;                       // S+ = P + iQ
    s.rm = pr + qi; s.im = pi_ - qr;                       // S- = P - iQ
}
__device__ __forceinline__ void cq_wave_sum(c3 &s) {
    s.r0 = wave_sum(s.r0); s.i0 = wave_sum(s.i0); s.rp = wave_sum(s.rp);
    s.ip = wave_sum(s.ip); s.rm = wave_sum(s.rm); s.im = wave_sum(s.im);
}
__device__ __forceinline__ void cq_add(c3 &s, const c3 &t) {
    s.r0 += t.r0; s.i0 += t.i0; s.rp += t.rp; s.ip += t.ip; s.rm += t.rm; s.im += t.im;
}

__global__ __launch_bounds__(256) void cqt_slices_kernel(amt_cqt_args a) {
    __shared__ float red[4][2 * AMT_CQT_MAXF];
    __shared__ c3 edgeP[CQ_EDGE], edgeQ[CQ_EDGE], coreS[4];
    const int k = blockIdx.x;                       // output bin
    const int b = blockIdx.y;
    const int kt = (a.bin0 ? a.bin0[b] : 0) + k;    // table row
    const int tid = threadIdx.x;
    float *o = a.out + ((size_t)b * a.n_bins + k) * a.frames;
    if (kt < 0 || kt >= a.n_table) {                // uniform; outside the table -> zeros
        if (tid < a.frames) o[tid] = 0.f;
        return;
    }
    const int nk = a.length[kt];
    const unsigned int inc = a.phase_inc[kt];
    const float *x = a.wave + (size_t)b * a.wave_stride;
    const int half = nk >> 1;
    const int H = a.hop;
    const int wid = tid >> 6, lane = tid & 63;

    int start[AMT_CQT_MAXF];
    int a_min = 0x7fffffff, a_max = -0x7fffffff;
    bool have = false;
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) {
        const int t = j < a.frames ? a.src_frame[b * a.frames + j] : -1;
        if (t < 0) { start[j] = 0x40000000; continue; }
        start[j] = t * H - half;
        have = true;
        a_min = min(a_min, start[j]);
        a_max = max(a_max, start[j]);
    }
    if (!have) { if (tid < a.frames) o[tid] = 0.f; return; }
    const float inv_nk = 1.0f / (float)nk;
    const float scale = 2.0f / sqrtf((float)nk);

    if (a_max - a_min <= 7 * H) {
        // ------------------------------ compact frame set -------------------------------
        const int nb = nk / H;                       // full blocks per frame
        const int rpre = nk - nb * H;                // prefix of the block behind a frame
        const int omax = (a_max - a_min) / H;
        const int nblk = omax + nb + 1;              // blocks 0 .. omax+nb
        if (tid < CQ_EDGE) { edgeP[tid] = c3{0, 0, 0, 0, 0, 0}; edgeQ[tid] = c3{0, 0, 0, 0, 0, 0}; }
        __syncthreads();
        c3 core{0, 0, 0, 0, 0, 0};
        for (int blk = wid; blk < nblk; blk += 4) {  // one wave per H-sample block
            const bool edge = blk < 7 || blk >= nb;  // wave-uniform
            c3 p{0, 0, 0, 0, 0, 0}, q{0, 0, 0, 0, 0, 0};
            const int mb = a_min + blk * H;                      // first sample of the block
            if (!edge && mb >= 0 && mb + H <= a.L) {
                // interior core block (the bulk of the work): no per-sample bounds test, no branches
                const float *xb = x + mb;
                auto body = [&](int i) {
                    const float xv = xb[i];
                    const float turns = (float)((unsigned int)(mb + i) * inc) * 2.3283064365386963e-10f;
                    const float sn = __builtin_amdgcn_sinf(turns), c = __builtin_amdgcn_cosf(turns);
                    const float wt = (float)(blk * H + i) * inv_nk;
                    const float sw = __builtin_amdgcn_sinf(wt), cw = __builtin_amdgcn_cosf(wt);
                    cq_accum(core, xv, c, sn, cw, sw);
                };
                if (H == 512) {                                  // the hop of the N = 2048 path: fully unrolled
#pragma unroll
                    for (int j = 0; j < 8; ++j) body(lane + 64 * j);
                } else {
                    for (int i = lane; i < H; i += 64) body(i);
                }
                continue;
            }
            for (int i = lane; i < H; i += 64) {
                const int u = blk * H + i;
                const int m = a_min + u;
                const float xv = (m >= 0 && m < a.L) ? x[m] : 0.f;
                // v_sin_f32 / v_cos_f32 take their argument in turns (1.0 = 2 pi), |x| <= 256
                const float turns = (float)((unsigned int)m * inc) * 2.3283064365386963e-10f;
                const float sn = __builtin_amdgcn_sinf(turns), c = __builtin_amdgcn_cosf(turns);
                const float wt = (float)u * inv_nk;
                const float sw = __builtin_amdgcn_sinf(wt), cw = __builtin_amdgcn_cosf(wt);
                if (edge) {
                    cq_accum(p, xv, c, sn, cw, sw);
                    if (i < rpre) cq_accum(q, xv, c, sn, cw, sw);
                } else {
                    cq_accum(core, xv, c, sn, cw, sw);
                }
            }
            if (edge) {
                const int e = blk < 7 ? blk : 7 + (blk - nb);
                cq_wave_sum(p);
                cq_wave_sum(q);
                if (lane == 0 && e < CQ_EDGE) { edgeP[e] = p; edgeQ[e] = q; }
            }
        }
        cq_wave_sum(core);
        if (lane == 0) coreS[wid] = core;
        __syncthreads();
        if (tid < a.frames) {
            float v = 0.f;
            const int st = a.src_frame[b * a.frames + tid];
            if (st >= 0) {
                const int oj = (st * H - half - a_min) / H;
                c3 s{0, 0, 0, 0, 0, 0};
                cq_add(s, coreS[0]); cq_add(s, coreS[1]); cq_add(s, coreS[2]); cq_add(s, coreS[3]);
                for (int blk = oj; blk < oj + nb; ++blk)
                    if (blk < 7 || blk >= nb) cq_add(s, edgeP[blk < 7 ? blk : 7 + (blk - nb)]);
                {
                    const int blk = oj + nb;         // the block behind the frame: its N_k mod H prefix
                    cq_add(s, edgeQ[blk < 7 ? blk : 7 + (blk - nb)]);
                }
                cq_finish(s);
                // C = 1/2 S0 - 1/4 (e^{-i theta o H} S+ + e^{+i theta o H} S-)
                float sd, cd;
                sincospif(2.0f * (float)(oj * H) * inv_nk, &sd, &cd);
                const float pr = cd * s.rp + sd * s.ip, pi_ = cd * s.ip - sd * s.rp;   // e^{-i d} S+
                const float mr = cd * s.rm - sd * s.im, mi = cd * s.im + sd * s.rm;   // e^{+i d} S-
                const float re = 0.5f * s.r0 - 0.25f * (pr + mr);
                const float im = 0.5f * s.i0 - 0.25f * (pi_ + mi);
                v = sqrtf(re * re + im * im) * scale;
                if (a.ref) v = __fdiv_rn(v, a.ref[b]);
            }
            o[tid] = v;
        }
        return;
    }

    // ------------------------------ generic frame set -----------------------------------
    float cd[AMT_CQT_MAXF], sd[AMT_CQT_MAXF];
    const int a0 = a_min;
    int m_lo = a_min, m_hi = a_max + nk;
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) {
        // cos(theta*(u - d)) = cos(theta u) cos(theta d) + sin(theta u) sin(theta d),
        // u = m - a0, d = start[j] - a0, theta = 2 pi / N_k   (angles in half-turns)
        const float d = (float)(start[j] - a0);
        sincospif(2.0f * d * inv_nk, &sd[j], &cd[j]);
        if (start[j] == 0x40000000) { cd[j] = 0.f; sd[j] = 0.f; }
    }
    m_lo = max(m_lo, 0);
    m_hi = min(m_hi, a.L);
    float re[AMT_CQT_MAXF], im[AMT_CQT_MAXF];
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) { re[j] = 0.f; im[j] = 0.f; }
    for (int m = m_lo + tid; m < m_hi; m += 256) {
        const float xv = x[m];
        const float turns = (float)((unsigned int)m * inc) * 2.3283064365386963e-10f;
        const float s = __builtin_amdgcn_sinf(turns), c = __builtin_amdgcn_cosf(turns);
        const float xr = xv * c, xi = -xv * s;                          // x * exp(-i phi)
        const float wt = (float)(m - a0) * inv_nk;
        const float sw = __builtin_amdgcn_sinf(wt), cw = __builtin_amdgcn_cosf(wt);
#pragma unroll
        for (int j = 0; j < AMT_CQT_MAXF; ++j) {
            const int n = m - start[j];
            float w = 0.5f - 0.5f * (cw * cd[j] + sw * sd[j]);
            w = (n >= 0 && n < nk) ? w : 0.f;
            re[j] += xr * w;
            im[j] += xi * w;
        }
    }
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) {
        const float r = wave_sum(re[j]), i = wave_sum(im[j]);
        if (lane == 0) { red[wid][2 * j] = r; red[wid][2 * j + 1] = i; }
    }
    __syncthreads();
    if (tid < a.frames) {
        const float r = red[0][2 * tid] + red[1][2 * tid] + red[2][2 * tid] + red[3][2 * tid];
        const float i = red[0][2 * tid + 1] + red[1][2 * tid + 1] + red[2][2 * tid + 1] + red[3][2 * tid + 1];
        float v = sqrtf(r * r + i * i) * scale;
        if (a.ref) v = __fdiv_rn(v, a.ref[b]);
        if (a.src_frame[b * a.frames + tid] < 0) v = 0.f;
        o[tid] = v;
    }
}

extern "C" int amt_cqt_slices(const amt_cqt_args *args, void *stream) {
    if (!args || !args->wave || !args->src_frame || !args->phase_inc || !args->length || !args->out)
        return AMT_E_INVALID;
    const amt_cqt_args &a = *args;
    if (a.B <= 0 || a.L <= 0 || a.hop <= 0 || a.n_bins <= 0 || a.n_table <= 0) return AMT_E_INVALID;
    if (a.frames <= 0 || a.frames > AMT_CQT_MAXF) return AMT_E_UNSUPPORTED;
    if (a.wave_stride < (size_t)a.L) return AMT_E_SHAPE;
    cqt_slices_kernel<<<dim3(a.n_bins, a.B), 256, 0, (hipStream_t)stream>>>(a);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---------------------------------------------------------------------------------------------
// Song-level normalisers: max over every bin and EVERY frame of a window's CQT
// (np.max(mid_wf.slice_C(0, duration, n_frames, ...)), /root/reference/training.py:271-282).
//
// One workgroup per (bin, window); the cost per bin is O(L) whatever N_k is.  With block j =
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

__global__ __launch_bounds__(256) void cqt_window_max_kernel(const float *__restrict__ wave, int L, size_t wave_stride,
                                                             int H, int hshift, int T,
                                                             const unsigned int *__restrict__ phase_inc,
                                                             const int *__restrict__ length,
                                                             const float *__restrict__ coef, int blk_cap,
                                                             unsigned int *__restrict__ out) {
    extern __shared__ double cm_smem[];
    float *stage = (float *)cm_smem;                 // [4][CM_STAGE]; the epilogue reuses it for PF
    double *pf = cm_smem;                            // [nblk + 1][6]
    float *fg = stage + max(4 * CM_STAGE, (blk_cap + 1) * 12);          // [blk_cap][12]  (F_j, G_j)
    const int k = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int nk = __builtin_amdgcn_readfirstlane(length[k]);
    const unsigned int inc = __builtin_amdgcn_readfirstlane(phase_inc[k]);
    const amt_v2 *__restrict__ cf = (const amt_v2 *)(coef + (size_t)k * CM_COEF);
    const float *x = wave + (size_t)b * wave_stride;
    const int half = nk >> 1;
    const int nb = nk >> hshift, rpre = nk - (nb << hshift);
    const int LPB = H >> 5;                          // lanes per block
    const int BPW = 64 / LPB, NBP = 4 * BPW;         // blocks per wave / per pass of the workgroup
    const int j_lo = half >> hshift, j_hi = (L - 1 + half) >> hshift;    // blocks that hold samples
    const int nblk = j_hi - j_lo + 1;                // <= blk_cap
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
    fetch(0);
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
    float vmax = 0.f;
    const float scale = 2.0f / sqrtf((float)nk);
    for (int t = tid; t < T; t += 256) {             // frame t: blocks t .. t+nb-1 whole, the head of block t+nb
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
        vmax = fmaxf(vmax, sqrtf(re * re + im * im) * scale);
    }
    vmax = wave_max(vmax);
    if (lane == 0) atomicMax(out + b, __float_as_uint(vmax));         // non-negative floats order as their bits
}

extern "C" int amt_cqt_window_max(const float *wave, int B, int L, size_t wave_stride, int hop,
                                  const uint32_t *phase_inc, const int32_t *length, int n_bins,
                                  float *coef_ws, float *out_max, void *stream) {
    if (!wave || !phase_inc || !length || !coef_ws || !out_max) return AMT_E_INVALID;
    if (B <= 0 || L <= 0 || n_bins <= 0) return AMT_E_INVALID;
    if (wave_stride < (size_t)L) return AMT_E_SHAPE;
    if (!amt_is_pow2(hop) || hop < 128 || hop > 2048) return AMT_E_UNSUPPORTED;
    int hshift = 0;
    while ((1 << hshift) < hop) ++hshift;
    const int blk_cap = L / hop + 3;                 // blocks that can hold samples, whatever the filter length
    const size_t stage_floats = (size_t)4 * CM_STAGE > (size_t)(blk_cap + 1) * 12 ? (size_t)4 * CM_STAGE
                                                                                   : (size_t)(blk_cap + 1) * 12;
    const size_t lds = (stage_floats + (size_t)blk_cap * 12) * sizeof(float);
    if (lds > 159 * 1024) return AMT_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        AMT_HIP_CHECK(hipFuncSetAttribute((const void *)cqt_window_max_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_set = true;
    }
    AMT_HIP_CHECK(hipMemsetAsync(out_max, 0, (size_t)B * sizeof(float), st));
    cqt_coef_kernel<<<(n_bins * 32 + 255) / 256, 256, 0, st>>>(phase_inc, length, n_bins, coef_ws);
    AMT_LAUNCH_CHECK();
    cqt_window_max_kernel<<<dim3(n_bins, B), 256, lds, st>>>(wave, L, wave_stride, hop, hshift, 1 + L / hop, phase_inc,
                                                            length, coef_ws, blk_cap, (unsigned int *)out_max);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}
