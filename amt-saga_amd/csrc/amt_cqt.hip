// Constant-Q slices at the (<= 8) frames the heads keep, for gfx950.
//
// Stands where audio_complete.slice_C calls |librosa.cqt| and keeps
// _resize(C[:, s:t], target) (/root/reference/util_audio.py:411-434).  The
// transform is the build's own spec (oracle/cqt.py): direct constant-Q
// response with periodic-Hann, L1-normalised filters of length N_k, scaled by
// sqrt(N_k); frequencies quantised to uint32 cycles/sample so the oscillator
// phase is exact integer arithmetic on CPU and GPU alike.
//
// One workgroup per (window, bin).  Every sample x[m] in the union of the
// frames' supports is read once (coalesced), multiplied by exp(-2 pi i phi(m))
// (one sincos, shared by all frames -- the per-frame phase offset has unit
// modulus and drops out of |.|), and added into each frame's accumulator with
// that frame's Hann weight, obtained from one more sincos by the angle-sum
// identity with per-frame constants.  VALU/L2-bound; no LDS tiles needed.
#include "amt_common.h"

#define AMT_CQT_MAXF 8

__global__ __launch_bounds__(256) void cqt_slices_kernel(amt_cqt_args a) {
    __shared__ float red[4][2 * AMT_CQT_MAXF];
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

    // per-frame window start a_j (absolute sample), Hann rotation constants
    int start[AMT_CQT_MAXF];
    float cd[AMT_CQT_MAXF], sd[AMT_CQT_MAXF];
    int m_lo = 0x7fffffff, m_hi = -0x7fffffff;
    int a0 = 0;
    bool have = false;
    const float inv_nk = 1.0f / (float)nk;
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) {
        const int t = j < a.frames ? a.src_frame[b * a.frames + j] : -1;
        if (t < 0) { start[j] = 0x40000000; cd[j] = 0.f; sd[j] = 0.f; continue; }
        start[j] = t * a.hop - half;
        if (!have) { a0 = start[j]; have = true; }
        m_lo = min(m_lo, start[j]);
        m_hi = max(m_hi, start[j] + nk);
    }
    if (!have) { if (tid < a.frames) o[tid] = 0.f; return; }
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
        const unsigned int ph = (unsigned int)m * inc;                 // exact mod 2^32
        // v_sin_f32 / v_cos_f32 take their argument in turns (1.0 = 2 pi), |x| <= 256
        const float turns = (float)ph * 2.3283064365386963e-10f;        // ph / 2^32
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
    const int wid = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < AMT_CQT_MAXF; ++j) {
        const float r = wave_sum(re[j]), i = wave_sum(im[j]);
        if (lane == 0) { red[wid][2 * j] = r; red[wid][2 * j + 1] = i; }
    }
    __syncthreads();
    if (tid < a.frames) {
        const float r = red[0][2 * tid] + red[1][2 * tid] + red[2][2 * tid] + red[3][2 * tid];
        const float i = red[0][2 * tid + 1] + red[1][2 * tid + 1] + red[2][2 * tid + 1] + red[3][2 * tid + 1];
        float v = sqrtf(r * r + i * i) * 2.0f / sqrtf((float)nk);
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
