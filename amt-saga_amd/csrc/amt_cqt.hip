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
