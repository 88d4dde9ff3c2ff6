// STFT -> |.| -> unit phase -> per-window max, and iSTFT, for gfx950.
//
// Replaces librosa.stft / librosa.core.magphase / np.max / librosa.istft as
// reached from audio_complete.F / .mag / .ph / .ref_mag / .wf
// (/root/reference/util_audio.py:88-174).
//
// Design (HBM-bound path; no MFMA on purpose):
//  * two real frames ride one complex FFT: z = w*(x_t + i*x_{t+1}); the two
//    spectra are separated by X_t[k] = (Z[k]+conj Z[N-k])/2,
//    X_{t+1}[k] = (Z[k]-conj Z[N-k])/(2i).
//  * a 256-thread workgroup owns `pairs_per_block` consecutive frame pairs of
//    one window; the N-point FFT runs in LDS (amt_fft.h) with LDS-staged
//    twiddles; the Hann window is derived from the same table
//    (w[n] = 0.5 - 0.5*Re W[n]).
//  * samples are read straight from global memory, consecutive threads ->
//    consecutive samples; the librosa reflect padding is an index map, never a
//    copy.  The 75 % frame overlap: the 2048 / hop 512 magnitude-only form keeps
//    a thread's ten samples of a frame pair in registers and loads only the four
//    new ones for the next pair (round 4); every other form re-reads through L2.
//  * spectra are written frame-major [t][f]: consecutive threads -> consecutive
//    bins, so every store instruction covers whole 128-B lines.
//  * |X| max is reduced wave -> block -> one atomicMax per block (non-negative
//    floats order like their bit patterns).
#include "amt_fft.h"

struct amt_stft_plan {
    int n_fft, hop, center;
    float2 *tw_dev;   // [n_fft] exp(-2 pi i m / n_fft)
};

thread_local char amt_hip_err_buf[256] = {0};

// (the magnitude-only 2048-point form: 80 registers with the ten carried samples -- six workgroups per CU, four spilled
// registers; five workgroups without spills measured 1.5 % slower, eight with thirteen spills did not fit the carry;
// round 3's form without the carry fitted 64 registers and eight workgroups and was 1.5-4 % slower at 9 % more HBM reads)
// Measured negative: the inter-stage twiddles are per-thread constants too (butterfly index = thread index), but
// holding all of them costs 40 registers -> 138, three waves per SIMD: 1.43 ms against 1.20; capped at 128 / 96 registers
// the compiler spills 8 / 37 of them.
template <int N, bool WITH_PHASE>
__global__ __launch_bounds__(AMT_FFT_THREADS, (N == 2048 && !WITH_PHASE) ? 6 : 1) void stft_mag_kernel(
    const float *__restrict__ wave, int L, size_t wave_stride,
    float *__restrict__ mag, float2 *__restrict__ phase, float *__restrict__ ref_max,
    int T, int ldf, size_t spec_stride, const float2 *__restrict__ tw_global,
    int hop, int center, int pairs_per_block, int reuse_ok) {
    // the first pass of fft_block reads elements tid + r * (N / 8) (radix 8, one butterfly per thread for N = 2048):
    // the same eight positions of every frame pair, so their Hann weights are formed once per workgroup, straight from
    // the global table -- and the LDS copy then only needs the entries the inter-stage twiddles touch: k * N / (NS R)
    // with k < NS, i.e. indices below N / 4 (the last pass is radix 4).  20 KB of LDS instead of 32: six workgroups per
    // CU (the registers' limit) instead of five.
    constexpr int NB1 = N / 8;
    constexpr bool HOIST = NB1 == AMT_FFT_THREADS;
    constexpr int TWN = HOIST ? N / 4 : N;
    __shared__ float2 buf[N];
    __shared__ float2 tw[TWN];
    // the reduction scratch aliases the FFT buffer (used once, after the last transform)
    float *red = reinterpret_cast<float *>(buf);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < TWN; i += AMT_FFT_THREADS) tw[i] = tw_global[i];
    float wn[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wn[r] = HOIST ? 0.25f - 0.25f * tw_global[tid + r * NB1].x : 0.f;   // Hann / 2 (below)
    __syncthreads();

    const float *wv = wave + (size_t)b * wave_stride;
    float *mg = mag + (size_t)b * spec_stride;
    float2 *ph = WITH_PHASE ? phase + (size_t)b * spec_stride : nullptr;
    const int pad = center ? N / 2 : 0;
    float lmax = 0.f;
    // Interior pairs of the hoisted form (N = 8 x 256 threads) with hop = N / 4: thread `tid` needs samples s0 + tid + 256 j,
    // j = 0 .. 7 for the first frame and j = 2 .. 9 for the second, and the next pair (s0 + 2 hops = s0 + 4 x 256) needs
    // j = 4 .. 13 of the same sequence: six of its ten samples are already in this thread's registers.  Kept there, a pair
    // costs 4 loads per thread instead of 16 and the 75 % frame overlap no longer goes to L2 at all (round 3: 1.54 MB read
    // per window against 1.055 of audio, the re-reads missing L2 under the 2 GB stream of magnitudes).
    constexpr bool REUSE = HOIST && !WITH_PHASE;
    const bool reuse = REUSE && reuse_ok && hop == 2 * AMT_FFT_THREADS;
    float sw[10];
    bool have = false;                                   // (uniform) sw holds the previous pair's samples
#pragma unroll
    for (int j = 0; j < 10; ++j) sw[j] = 0.f;
    for (int p = 0; p < pairs_per_block; ++p) {
        const int t0 = 2 * (blockIdx.x * pairs_per_block + p);
        if (t0 >= T) break;                       // uniform across the block
        const bool has2 = (t0 + 1) < T;
        const int s0 = t0 * hop - pad;            // first sample of frame t0
        // interior pairs (both frames inside the signal) skip the reflect index map
        const bool interior = s0 >= 0 && (s0 + hop + N) <= L && has2;
        auto load_edge = [&](int n) -> float2 {
            const float w = HOIST ? wn[(n - tid) / NB1] : 0.25f - 0.25f * tw[n].x;
            int i0 = s0 + n;
            int i1 = i0 + hop;
            // numpy 'reflect' (single reflection; pad = N/2 <= L-1 checked on host)
            i0 = i0 < 0 ? -i0 : i0;  i0 = i0 >= L ? 2 * (L - 1) - i0 : i0;
            i1 = i1 < 0 ? -i1 : i1;  i1 = i1 >= L ? 2 * (L - 1) - i1 : i1;
            const float a = wv[i0];
            const float c = has2 ? wv[i1] : 0.f;
            return make_float2(a * w, c * w);
        };
        const float *w0 = wv + s0;
        auto load_in = [&](int n) -> float2 {
            const float w = HOIST ? wn[(n - tid) / NB1] : 0.25f - 0.25f * tw[n].x;
            return make_float2(w0[n] * w, w0[n + hop] * w);
        };
        if (REUSE && reuse && interior) {
            if (have) {
#pragma unroll
                for (int j = 0; j < 6; ++j) sw[j] = sw[j + 4];
#pragma unroll
                for (int j = 6; j < 10; ++j) sw[j] = w0[tid + AMT_FFT_THREADS * j];
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) sw[j] = w0[tid + AMT_FFT_THREADS * j];
            }
            have = true;
            auto load_regs = [&](int n) -> float2 {
                const int r = (n - tid) / NB1;            // (compile-time after unrolling, as for wn[])
                return make_float2(sw[r] * wn[r], sw[r + 2] * wn[r]);
            };
            fft_block<N, false>(buf, tw, load_regs);
        } else {
            have = false;
            if (interior) fft_block<N, false>(buf, tw, load_in);
            else fft_block<N, false>(buf, tw, load_edge);
        }

        // The window above is Hann / 2 (a power-of-two factor: exact), which is the 1/2 of the separation
        //   X_t[k] = (Z[k] + conj Z[N-k]) / 2,  X_{t+1}[k] = (Z[k] - conj Z[N-k]) / (2i).
        float *m0 = mg + (size_t)t0 * ldf;
        float *m1 = m0 + ldf;
        if constexpr (!WITH_PHASE) {
            // magnitudes only: bins 0 .. N/2 - 1 in whole passes of the workgroup (no guard, v_sqrt_f32 instead of the
            // rsq / compare / select / multiply the phase form needs), then bin N/2 and the pad columns by the first lanes
            for (int k = tid; k < N / 2; k += AMT_FFT_THREADS) {
                const float2 zk = buf[k];
                const float2 zm = cconj(buf[(N - k) & (N - 1)]);
                const float2 x0 = cadd(zk, zm);
                const float2 d = csub(zk, zm);
                const float a0 = __builtin_amdgcn_sqrtf(x0.x * x0.x + x0.y * x0.y);
                const float a1 = __builtin_amdgcn_sqrtf(d.y * d.y + d.x * d.x);
                lmax = fmaxf(lmax, a0);
                // streaming stores: the 2 GB of magnitudes are next read by other kernels, long after the L2 has turned
                // over; kept out of it, the L2 holds the samples the NEXT frame pair re-reads (75 % overlap)
                __builtin_nontemporal_store(a0, m0 + k);
                if (has2) { lmax = fmaxf(lmax, a1); __builtin_nontemporal_store(a1, m1 + k); }
            }
            for (int k = N / 2 + tid; k < ldf; k += AMT_FFT_THREADS) {     // ldf - N/2 = 4 columns: the first wave only
                float a0 = 0.f, a1 = 0.f;
                if (k == N / 2) {                                // Z[N/2] = X_t[N/2] + i X_{t+1}[N/2], both real
                    const float2 zk = buf[N / 2];
                    const float xr = zk.x + zk.x, xi = zk.y + zk.y;
                    a0 = __builtin_amdgcn_sqrtf(xr * xr);
                    a1 = __builtin_amdgcn_sqrtf(xi * xi);
                    lmax = fmaxf(lmax, a0);
                    if (has2) lmax = fmaxf(lmax, a1);
                }
                m0[k] = a0;
                if (has2) m1[k] = a1;
            }
        } else {
        for (int k = tid; k < ldf; k += AMT_FFT_THREADS) {
            float a0 = 0.f, a1 = 0.f;
            float2 p0 = make_float2(0.f, 0.f), p1 = p0;
            if (k <= N / 2) {
                const float2 zk = buf[k];
                const float2 zm = cconj(buf[(N - k) & (N - 1)]);
                const float2 x0 = cadd(zk, zm);
                const float2 d = csub(zk, zm);
                const float2 x1 = make_float2(d.y, -d.x);
                // |X| = n2 * rsq(n2), X/|X| = X * rsq(n2)  (v_rsq_f32, ~1 ulp); 0 -> (0, 1+0i)
                const float n0 = x0.x * x0.x + x0.y * x0.y;
                const float n1 = x1.x * x1.x + x1.y * x1.y;
                const float r0 = n0 > 1e-36f ? __builtin_amdgcn_rsqf(n0) : 0.f;
                const float r1 = n1 > 1e-36f ? __builtin_amdgcn_rsqf(n1) : 0.f;
                a0 = n0 * r0;
                a1 = n1 * r1;
                p0 = r0 > 0.f ? make_float2(x0.x * r0, x0.y * r0) : make_float2(1.f, 0.f);
                p1 = r1 > 0.f ? make_float2(x1.x * r1, x1.y * r1) : make_float2(1.f, 0.f);
                lmax = fmaxf(lmax, a0);
                if (has2) lmax = fmaxf(lmax, a1);
            }
            m0[k] = a0;
            if (has2) m1[k] = a1;
            ph[(size_t)t0 * ldf + k] = p0;
            if (has2) ph[(size_t)(t0 + 1) * ldf + k] = p1;
        }
        }
        // the next pair's first pass barriers before it overwrites `buf`
    }
    if (ref_max) {
        __syncthreads();                                 // every wave is done with `buf`
        lmax = block_max(lmax, red);
        if (tid == 0) atomicMax(reinterpret_cast<int *>(ref_max) + b, __float_as_int(lmax));
    }
}

// ---------------------------------------------------------------------------------
// iSTFT: deterministic gather (no float atomics).  A workgroup owns GH output
// hops; it inverse-transforms every frame overlapping them (pairs again),
// accumulates the Hann-windowed frames into an LDS segment, divides by the
// window sum-of-squares of the frames that exist, trims n_fft/2 (center) and
// writes the samples coalesced.
// ---------------------------------------------------------------------------------
template <int N, int GH_HOP_MAX>
__global__ __launch_bounds__(AMT_FFT_THREADS) void istft_kernel(
    const float *__restrict__ mag, const float2 *__restrict__ phase, int T, int ldf,
    size_t spec_stride, float *__restrict__ wave_out, size_t wave_stride, int Lout,
    const float2 *__restrict__ tw_global, int hop, int center, int GH) {
    __shared__ float2 buf[N];
    __shared__ float2 tw[N];
    __shared__ float acc[GH_HOP_MAX];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < N; i += AMT_FFT_THREADS) tw[i] = tw_global[i];
    const int seg = GH * hop;                       // padded samples owned by this block
    for (int i = tid; i < seg; i += AMT_FFT_THREADS) acc[i] = 0.f;
    __syncthreads();

    const int pad = center ? N / 2 : 0;
    const int p_lo = blockIdx.x * seg;              // first padded sample of the segment
    const int p_hi = p_lo + seg;
    // frames t with [t*hop, t*hop+N) intersecting [p_lo, p_hi)
    int t_lo = (p_lo - N + hop) / hop;              // ceil((p_lo-N+1)/hop) for p_lo-N+1 > 0
    if (p_lo - N + 1 <= 0) t_lo = 0;
    int t_hi = (p_hi - 1) / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    const float *mg = mag + (size_t)b * spec_stride;
    const float2 *ph = phase ? phase + (size_t)b * spec_stride : nullptr;
    const float inv_n = 1.0f / (float)N;

    for (int t0 = t_lo; t0 <= t_hi; t0 += 2) {
        const bool has2 = (t0 + 1) <= t_hi;
        auto spec = [&](int t, int k) -> float2 {
            float2 v;
            if (ph) {
                const float m = mg[(size_t)t * ldf + k];
                const float2 q = ph[(size_t)t * ldf + k];
                v = make_float2(m * q.x, m * q.y);
            } else {
                v = reinterpret_cast<const float2 *>(mg)[(size_t)t * ldf + k];
            }
            if (k == 0 || k == N / 2) v.y = 0.f;   // C2R ignores these imaginary parts
            return v;
        };
        auto load = [&](int n) -> float2 {
            const bool mirror = n > N / 2;
            const int k = mirror ? N - n : n;
            float2 x1 = spec(t0, k);
            float2 x2 = has2 ? spec(t0 + 1, k) : make_float2(0.f, 0.f);
            if (mirror) { x1.y = -x1.y; x2.y = -x2.y; }
            return make_float2(x1.x - x2.y, x1.y + x2.x);   // x1 + i*x2
        };
        fft_block<N, true>(buf, tw, load);
        // overlap-add frame t0 then t0+1 (distinct n per thread -> no LDS races)
        for (int f = 0; f < 2; ++f) {
            if (f == 1 && !has2) break;
            const int base = (t0 + f) * hop - p_lo;
            for (int n = tid; n < N; n += AMT_FFT_THREADS) {
                const int loc = base + n;
                if (loc >= 0 && loc < seg) {
                    const float w = 0.5f - 0.5f * tw[n].x;
                    const float v = (f == 0 ? buf[n].x : buf[n].y) * inv_n * w;
                    acc[loc] += v;
                }
            }
            __syncthreads();
        }
    }
    float *out = wave_out + (size_t)b * wave_stride;
    for (int i = tid; i < seg; i += AMT_FFT_THREADS) {
        const int p = p_lo + i;
        const int o = p - pad;
        if (o < 0 || o >= Lout) continue;
        // window sum-of-squares over the frames that exist (librosa window_sumsquare)
        int ta = (p - N + hop) / hop;
        if (p - N + 1 <= 0) ta = 0;
        int tb = p / hop;
        if (tb > T - 1) tb = T - 1;
        float wss = 0.f;
        for (int t = ta; t <= tb; ++t) {
            const float w = 0.5f - 0.5f * tw[p - t * hop].x;
            wss += w * w;
        }
        float v = acc[i];
        if (wss > 1.17549435e-38f) v /= wss;
        out[o] = v;
    }
}

// ---------------------------------------------------------------------------------
// iSTFT, streaming form for hop = N/4 (librosa's default, every configuration of the path).
// With N/hop = 4 and hop a multiple of the 256 threads, thread `tid` is the only one that
// ever touches the segment samples = tid (mod 256), in every frame.  So the overlap-add needs
// neither an LDS accumulator nor barriers of its own: each thread keeps a sliding window of
// N/256 partial sums in registers; after frame slot s the hop/256 oldest are complete (no
// later frame reaches them), get divided by the window sum-of-squares and are stored
// (coalesced), and the window slides on.  LDS = FFT buffer + twiddles only (32 KB at N = 2048,
// five workgroups per CU instead of two); frames are visited in increasing t, so the sums are
// formed in the same order as in istft_kernel.
// ---------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(AMT_FFT_THREADS) void istft_stream_kernel(
    const float *__restrict__ mag, const float2 *__restrict__ phase, int T, int ldf,
    size_t spec_stride, float *__restrict__ wave_out, size_t wave_stride, int Lout,
    const float2 *__restrict__ tw_global, int center, int GH) {
    static_assert(AMT_FFT_THREADS == 256 && N % 1024 == 0, "hop = N/4 must be a multiple of the block");
    constexpr int HOP = N / 4, HPT = HOP / 256, NA = N / 256;
    __shared__ float2 buf[N];
    __shared__ float2 tw[N];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    for (int i = tid; i < N; i += AMT_FFT_THREADS) tw[i] = tw_global[i];
    __syncthreads();
    const int pad = center ? N / 2 : 0;
    const int seg = GH * HOP;
    const int p_lo = blockIdx.x * seg;
    const int t_base = p_lo / HOP - 3;               // frame slot 0: the first frame reaching p_lo
    const int nslots = GH + 3;
    const float *mg = mag + (size_t)b * spec_stride;
    const float2 *ph = phase ? phase + (size_t)b * spec_stride : nullptr;
    float *out = wave_out + (size_t)b * wave_stride;
    const float inv_n = 1.0f / (float)N;
    float a[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = 0.f;

    for (int s0 = 0; s0 < nslots; s0 += 2) {
        const int t0 = t_base + s0;
        const bool v0 = t0 >= 0 && t0 < T;
        const bool v1 = (s0 + 1 < nslots) && (t0 + 1) >= 0 && (t0 + 1) < T;
        if (v0 || v1) {
            auto spec = [&](int t, int k) -> float2 {
                float2 v;
                if (ph) {
                    const float m = mg[(size_t)t * ldf + k];
                    const float2 q = ph[(size_t)t * ldf + k];
                    v = make_float2(m * q.x, m * q.y);
                } else {
                    v = reinterpret_cast<const float2 *>(mg)[(size_t)t * ldf + k];
                }
                if (k == 0 || k == N / 2) v.y = 0.f;
                return v;
            };
            // the packed spectrum Z = X1 + i X2 and its Hermitian half are staged in LDS from ONE read of
            // every bin (a loader-per-element first pass would fetch each bin twice: direct and mirrored)
            __syncthreads();                                 // the previous pair's overlap-add is done with `buf`
            for (int k = tid; k <= N / 2; k += AMT_FFT_THREADS) {
                const float2 x1 = v0 ? spec(t0, k) : make_float2(0.f, 0.f);
                const float2 x2 = v1 ? spec(t0 + 1, k) : make_float2(0.f, 0.f);
                buf[k] = make_float2(x1.x - x2.y, x1.y + x2.x);                  // x1 + i x2
                if (k > 0 && k < N / 2) buf[N - k] = make_float2(x1.x + x2.y, x2.x - x1.y);   // conj(x1) + i conj(x2)
            }
            __syncthreads();
            fft_block_inplace<N, true>(buf, tw);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int s_ = s0 + f;
            if (s_ >= nslots) break;
            if (f == 0 ? v0 : v1) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int n = tid + 256 * i;
                    const float w = 0.5f - 0.5f * tw[n].x;
                    a[i] += (f == 0 ? buf[n].x : buf[n].y) * inv_n * w;
                }
            }
            // the HPT oldest sums are complete: segment samples tid + 256 * (HPT * (s - 3) + e)
#pragma unroll
            for (int e = 0; e < HPT; ++e) {
                const int j = HPT * (s_ - 3) + e;
                if (j < 0 || j >= GH * HPT) continue;
                const int p = p_lo + tid + 256 * j;
                const int o = p - pad;
                if (o < 0 || o >= Lout) continue;
                int ta = (p - N + HOP) / HOP;
                if (p - N + 1 <= 0) ta = 0;
                int tb = p / HOP;
                if (tb > T - 1) tb = T - 1;
                float wss = 0.f;
                for (int t = ta; t <= tb; ++t) {
                    const float w = 0.5f - 0.5f * tw[p - t * HOP].x;
                    wss += w * w;
                }
                float v = a[e];
                if (wss > 1.17549435e-38f) v /= wss;
                out[o] = v;
            }
#pragma unroll
            for (int i = 0; i < NA - HPT; ++i) a[i] = a[i + HPT];
#pragma unroll
            for (int i = NA - HPT; i < NA; ++i) a[i] = 0.f;
        }
    }
}

// np.max over a window's [T][ldf] block --------------------------------------------
__global__ __launch_bounds__(256) void window_max_kernel(const float *__restrict__ spec, int n,
                                                          size_t spec_stride,
                                                          unsigned int *__restrict__ out) {
    __shared__ float red[16];
    const float *s = spec + (size_t)blockIdx.y * spec_stride;
    float m = -INFINITY;
    const int n4 = n >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(s);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const float4 v = s4[i];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) atomicMax(out + blockIdx.y, float_to_ordered(m));
}
__global__ void ordered_init_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = AMT_ORDERED_NEG_INF;
}
__global__ void ordered_decode_kernel(unsigned int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __float_as_uint(ordered_to_float(p[i]));
}

// host launchers
template <int N>
static int launch_stft(const amt_stft_plan *plan, const float *wave, int B, int L,
                       size_t wave_stride, float *mag, float *phase, float *ref_max, int T,
                       int ldf, size_t spec_stride, hipStream_t st) {
    const int pairs = (T + 1) / 2;
    static int ppb_max = 0;                              // AMT_STFT_PPB: frame pairs per workgroup (diagnostic)
    if (!ppb_max) { const char *e = getenv("AMT_STFT_PPB"); ppb_max = (e && atoi(e) > 0) ? atoi(e) : 16; }
    int ppb = ppb_max;
    // at least ~8 rounds of the 2048 workgroups the chip holds (8 per CU): fewer, longer workgroups leave a tail
    while (ppb > 1 && (size_t)((pairs + ppb - 1) / ppb) * B < 16384) ppb >>= 1;
    dim3 grid((pairs + ppb - 1) / ppb, B);
    static int reuse_ok = -1;                            // AMT_STFT_REUSE=0: every pair loads its 16 samples per thread (diagnostic)
    if (reuse_ok < 0) { const char *e = getenv("AMT_STFT_REUSE"); reuse_ok = !(e && atoi(e) == 0); }
    if (phase)
        stft_mag_kernel<N, true><<<grid, AMT_FFT_THREADS, 0, st>>>(
            wave, L, wave_stride, mag, reinterpret_cast<float2 *>(phase), ref_max, T, ldf,
            spec_stride, plan->tw_dev, plan->hop, plan->center, ppb, reuse_ok);
    else
        stft_mag_kernel<N, false><<<grid, AMT_FFT_THREADS, 0, st>>>(
            wave, L, wave_stride, mag, nullptr, ref_max, T, ldf, spec_stride, plan->tw_dev,
            plan->hop, plan->center, ppb, reuse_ok);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

template <int N>
static int launch_istft(const amt_stft_plan *plan, const float *mag, const float *phase, int B,
                        int T, int ldf, size_t spec_stride, float *out, size_t wave_stride,
                        int Lout, hipStream_t st) {
    constexpr int GH_HOP_MAX = 8 * 1024;            // floats of LDS accumulator (32 KB)
    const int hop = plan->hop;
    if constexpr (N % 1024 == 0) {
        if (hop * 4 == N) {
            const int pad_ = plan->center ? N / 2 : 0;
            int GHs = 32;
            // keep >= ~2048 workgroups in flight when the batch is small
            while (GHs > 4 && (size_t)((pad_ + Lout + GHs * hop - 1) / (GHs * hop)) * B < 2048) GHs >>= 1;
            dim3 grid_s((pad_ + Lout + GHs * hop - 1) / (GHs * hop), B);
            istft_stream_kernel<N><<<grid_s, AMT_FFT_THREADS, 0, st>>>(
                mag, reinterpret_cast<const float2 *>(phase), T, ldf, spec_stride, out, wave_stride, Lout,
                plan->tw_dev, plan->center, GHs);
            AMT_LAUNCH_CHECK();
            return AMT_OK;
        }
    }
    int GH = GH_HOP_MAX / hop;
    if (GH < 1) return AMT_E_UNSUPPORTED;
    if (GH > 16) GH = 16;
    const int pad = plan->center ? N / 2 : 0;
    const int total_p = pad + Lout;                 // padded samples that matter
    dim3 grid((total_p + GH * hop - 1) / (GH * hop), B);
    istft_kernel<N, GH_HOP_MAX><<<grid, AMT_FFT_THREADS, 0, st>>>(
        mag, reinterpret_cast<const float2 *>(phase), T, ldf, spec_stride, out, wave_stride, Lout,
        plan->tw_dev, hop, plan->center, GH);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
extern "C" {

int amt_version(void) { return 1; }

const char *amt_strerror(int s) {
    switch (s) {
        case AMT_OK: return "ok";
        case AMT_E_INVALID: return "invalid argument";
        case AMT_E_SHAPE: return "Invalid Input shape";
        case AMT_E_HIP: return "HIP runtime error";
        case AMT_E_NOMEM: return "out of memory";
        case AMT_E_UNSUPPORTED: return "unsupported configuration";
        case AMT_E_ATTRIB: return "Requested attribute does not exist";
        default: return "unknown status";
    }
}

const char *amt_last_hip_error(void) { return amt_hip_err_buf; }

int amt_device_info(int *cu_count, int *lds_bytes, char *arch, int arch_len) {
    int dev = 0;
    AMT_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    AMT_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return AMT_OK;
}

int amt_stft_plan_create(amt_stft_plan **plan, int n_fft, int hop, int center) {
    if (!plan || !amt_is_pow2(n_fft) || n_fft < 256 || n_fft > 4096 || hop <= 0 || hop > n_fft)
        return AMT_E_INVALID;
    amt_stft_plan *p = new amt_stft_plan();
    p->n_fft = n_fft; p->hop = hop; p->center = center ? 1 : 0; p->tw_dev = nullptr;
    float2 *h = new float2[n_fft];
    for (int m = 0; m < n_fft; ++m) {
        const double a = -2.0 * 3.14159265358979323846 * (double)m / (double)n_fft;
        h[m] = make_float2((float)cos(a), (float)sin(a));
    }
    hipError_t e = hipMalloc(&p->tw_dev, sizeof(float2) * n_fft);
    if (e == hipSuccess) e = hipMemcpy(p->tw_dev, h, sizeof(float2) * n_fft, hipMemcpyHostToDevice);
    delete[] h;
    if (e != hipSuccess) {
        snprintf(amt_hip_err_buf, sizeof(amt_hip_err_buf), "plan upload: %s", hipGetErrorString(e));
        if (p->tw_dev) (void)hipFree(p->tw_dev);
        delete p;
        return AMT_E_HIP;
    }
    *plan = p;
    return AMT_OK;
}

int amt_stft_plan_destroy(amt_stft_plan *plan) {
    if (!plan) return AMT_OK;
    if (plan->tw_dev) (void)hipFree(plan->tw_dev);
    delete plan;
    return AMT_OK;
}

int amt_stft_frames(const amt_stft_plan *plan, int L) {
    if (!plan || L <= 0) return AMT_E_INVALID;
    if (plan->center) return 1 + L / plan->hop;
    if (L < plan->n_fft) return AMT_E_SHAPE;
    return 1 + (L - plan->n_fft) / plan->hop;
}

int amt_stft_mag(const amt_stft_plan *plan, const float *wave, int B, int L, size_t wave_stride,
                 float *mag, float *phase_ri, float *ref_max, int T, int ldf,
                 size_t spec_stride, void *stream) {
    if (!plan || !wave || !mag || B <= 0 || L <= 0) return AMT_E_INVALID;
    const int F = plan->n_fft / 2 + 1;
    if (ldf < F || (ldf & 3) || T != amt_stft_frames(plan, L)) return AMT_E_SHAPE;
    if (spec_stride < (size_t)T * ldf || wave_stride < (size_t)L) return AMT_E_SHAPE;
    if (plan->center && L <= plan->n_fft / 2) return AMT_E_SHAPE;   // reflect pad needs L > n_fft/2
    hipStream_t st = (hipStream_t)stream;
    if (ref_max) AMT_HIP_CHECK(hipMemsetAsync(ref_max, 0, sizeof(float) * B, st));
    switch (plan->n_fft) {
        case 256:  return launch_stft<256>(plan, wave, B, L, wave_stride, mag, phase_ri, ref_max, T, ldf, spec_stride, st);
        case 512:  return launch_stft<512>(plan, wave, B, L, wave_stride, mag, phase_ri, ref_max, T, ldf, spec_stride, st);
        case 1024: return launch_stft<1024>(plan, wave, B, L, wave_stride, mag, phase_ri, ref_max, T, ldf, spec_stride, st);
        case 2048: return launch_stft<2048>(plan, wave, B, L, wave_stride, mag, phase_ri, ref_max, T, ldf, spec_stride, st);
        case 4096: return launch_stft<4096>(plan, wave, B, L, wave_stride, mag, phase_ri, ref_max, T, ldf, spec_stride, st);
    }
    return AMT_E_UNSUPPORTED;
}

int amt_istft(const amt_stft_plan *plan, const float *mag, const float *phase_ri, int B, int T,
              int ldf, size_t spec_stride, float *wave_out, size_t wave_stride, void *stream) {
    if (!plan || !mag || !wave_out || B <= 0 || T <= 0) return AMT_E_INVALID;
    const int F = plan->n_fft / 2 + 1;
    if (ldf < F) return AMT_E_SHAPE;
    const int Lout = plan->center ? plan->hop * (T - 1) : plan->n_fft + plan->hop * (T - 1);
    if (Lout <= 0 || wave_stride < (size_t)Lout) return AMT_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    switch (plan->n_fft) {
        case 256:  return launch_istft<256>(plan, mag, phase_ri, B, T, ldf, spec_stride, wave_out, wave_stride, Lout, st);
        case 512:  return launch_istft<512>(plan, mag, phase_ri, B, T, ldf, spec_stride, wave_out, wave_stride, Lout, st);
        case 1024: return launch_istft<1024>(plan, mag, phase_ri, B, T, ldf, spec_stride, wave_out, wave_stride, Lout, st);
        case 2048: return launch_istft<2048>(plan, mag, phase_ri, B, T, ldf, spec_stride, wave_out, wave_stride, Lout, st);
        case 4096: return launch_istft<4096>(plan, mag, phase_ri, B, T, ldf, spec_stride, wave_out, wave_stride, Lout, st);
    }
    return AMT_E_UNSUPPORTED;
}

int amt_window_max(const float *spec, int B, int T, int ldf, size_t spec_stride, float *out_max,
                   void *stream) {
    if (!spec || !out_max || B <= 0 || T <= 0 || ldf <= 0 || (ldf & 3)) return AMT_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    unsigned int *o = reinterpret_cast<unsigned int *>(out_max);
    ordered_init_kernel<<<(B + 255) / 256, 256, 0, st>>>(o, B);
    const int n = T * ldf;
    int gx = (n / 4 + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    window_max_kernel<<<dim3(gx, B), 256, 0, st>>>(spec, n, spec_stride, o);
    ordered_decode_kernel<<<(B + 255) / 256, 256, 0, st>>>(o, B);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

}  // extern "C"
