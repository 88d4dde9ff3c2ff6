// Workgroup-cooperative in-place Stockham FFT in LDS for gfx950.
//
// One 256-thread workgroup transforms one N-point complex sequence held in an
// LDS buffer `buf[N]`; twiddles W[m] = exp(-2*pi*i*m/N) (computed in double on
// the host) are staged in LDS `tw[N]` -- "LDS-staged FFT twiddles".  Each pass
// of radix R: every thread pulls its butterflies' R inputs into registers,
// barrier, applies the inter-stage twiddles, does the R-point DFT in registers
// and scatters to the autosort positions, barrier.  No bit reversal.
//
//   pass with sub-transform size Ns (1, R1, R1*R2, ...), butterfly j in [0, N/R):
//     k    = j mod Ns
//     in   = buf[j + r*N/R],                r = 0..R-1
//     in_r *= W[r*k*N/(Ns*R)]
//     out  = DFT_R(in)
//     buf[(j-k)*R + k + r*Ns] = out_r
#pragma once
#include "amt_common.h"

#define AMT_FFT_THREADS 256

template <int R> struct Dft;

template <> struct Dft<2> {
    __device__ static __forceinline__ void run(float2 (&x)[2]) {
        float2 a = cadd(x[0], x[1]), b = csub(x[0], x[1]);
        x[0] = a; x[1] = b;
    }
};

template <> struct Dft<4> {
    // INV selects the conjugate transform
    template <bool INV>
    __device__ static __forceinline__ void run(float2 (&x)[4]) {
        float2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
        float2 s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
        float2 r = INV ? cmul_pi(d13) : cmul_mi(d13);
        x[0] = cadd(s02, s13);
        x[1] = cadd(d02, r);
        x[2] = csub(s02, s13);
        x[3] = csub(d02, r);
    }
};

template <> struct Dft<8> {
    template <bool INV>
    __device__ static __forceinline__ void run(float2 (&x)[8]) {
        const float h = 0.70710678118654752440f;
        float2 a0 = cadd(x[0], x[4]), a1 = csub(x[0], x[4]);
        float2 a2 = cadd(x[2], x[6]), t3 = csub(x[2], x[6]);
        float2 a4 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]);
        float2 a6 = cadd(x[3], x[7]), t7 = csub(x[3], x[7]);
        float2 a3 = INV ? cmul_pi(t3) : cmul_mi(t3);
        float2 a7 = INV ? cmul_pi(t7) : cmul_mi(t7);
        float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
        float2 b1 = cadd(a1, a3), b3 = csub(a1, a3);
        float2 b4 = cadd(a4, a6), b6 = csub(a4, a6);
        float2 b5 = cadd(a5, a7), b7 = csub(a5, a7);
        // w = exp(-+ i*pi/4): forward (h, -h), inverse (h, +h)
        float2 w1b5 = INV ? make_float2(h * (b5.x - b5.y), h * (b5.x + b5.y))
                          : make_float2(h * (b5.x + b5.y), h * (b5.y - b5.x));
        float2 w2b6 = INV ? cmul_pi(b6) : cmul_mi(b6);
        // w^3 = exp(-+ 3i*pi/4): forward (-h, -h), inverse (-h, +h)
        float2 w3b7 = INV ? make_float2(-h * (b7.x + b7.y), h * (b7.x - b7.y))
                          : make_float2(h * (b7.y - b7.x), -h * (b7.x + b7.y));
        x[0] = cadd(b0, b4);   x[4] = csub(b0, b4);
        x[1] = cadd(b1, w1b5); x[5] = csub(b1, w1b5);
        x[2] = cadd(b2, w2b6); x[6] = csub(b2, w2b6);
        x[3] = cadd(b3, w3b7); x[7] = csub(b3, w3b7);
    }
};

template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2 (&x)[R]) {
    if constexpr (R == 2) Dft<2>::run(x);
    else Dft<R>::template run<INV>(x);
}

// One in-place pass.  LOADER(n) supplies element n for the first pass (from
// global memory); later passes read `buf`.
template <int N, int R, int NS, bool INV, bool FIRST, typename Loader>
__device__ __forceinline__ void fft_pass(float2 *buf, const float2 *tw, Loader load) {
    constexpr int NB = N / R;                       // butterflies
    constexpr int BPT = (NB + AMT_FFT_THREADS - 1) / AMT_FFT_THREADS;
    static_assert(NB % AMT_FFT_THREADS == 0 || NB < AMT_FFT_THREADS, "butterfly split");
    float2 x[BPT][R];
    const int tid = threadIdx.x;
#pragma unroll
    for (int bi = 0; bi < BPT; ++bi) {
        const int j = tid + bi * AMT_FFT_THREADS;
        if (NB >= AMT_FFT_THREADS || j < NB) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr (FIRST) x[bi][r] = load(j + r * NB);
                else x[bi][r] = buf[j + r * NB];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int bi = 0; bi < BPT; ++bi) {
        const int j = tid + bi * AMT_FFT_THREADS;
        if (NB >= AMT_FFT_THREADS || j < NB) {
            const int k = j & (NS - 1);
            if constexpr (NS > 1) {
                // one LDS read; w^2..w^(R-1) by multiplication (error <= a few ulp, and no
                // bank-conflicted strided table walks)
                constexpr int TSTEP = N / (NS * R);
                float2 w1 = tw[k * TSTEP];
                if (INV) w1.y = -w1.y;
                float2 wp[R];
                wp[1] = w1;
                if constexpr (R > 2) wp[2] = cmul(w1, w1);
                if constexpr (R > 3) wp[3] = cmul(wp[2], w1);
                if constexpr (R > 4) {
                    wp[4] = cmul(wp[2], wp[2]);
                    wp[5] = cmul(wp[4], w1);
                    wp[6] = cmul(wp[3], wp[3]);
                    wp[7] = cmul(wp[4], wp[3]);
                }
#pragma unroll
                for (int r = 1; r < R; ++r) x[bi][r] = cmul(x[bi][r], wp[r]);
            }
            dft_r<R, INV>(x[bi]);
            const int base = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[base + r * NS] = x[bi][r];
        }
    }
    __syncthreads();
}

// Full transform for the supported sizes (radix schedule 8,8,...,{8|4|2}).
template <int N, bool INV, typename Loader>
__device__ __forceinline__ void fft_block(float2 *buf, const float2 *tw, Loader load) {
    auto none = [](int) { return make_float2(0.f, 0.f); };
    if constexpr (N == 4096) {
        fft_pass<N, 8, 1, INV, true>(buf, tw, load);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 8, 64, INV, false>(buf, tw, none);
        fft_pass<N, 8, 512, INV, false>(buf, tw, none);
    } else if constexpr (N == 2048) {
        fft_pass<N, 8, 1, INV, true>(buf, tw, load);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 8, 64, INV, false>(buf, tw, none);
        fft_pass<N, 4, 512, INV, false>(buf, tw, none);
    } else if constexpr (N == 1024) {
        fft_pass<N, 8, 1, INV, true>(buf, tw, load);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 4, 64, INV, false>(buf, tw, none);
        fft_pass<N, 4, 256, INV, false>(buf, tw, none);
    } else if constexpr (N == 512) {
        fft_pass<N, 8, 1, INV, true>(buf, tw, load);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 8, 64, INV, false>(buf, tw, none);
    } else {
        static_assert(N == 256, "unsupported FFT size");
        fft_pass<N, 8, 1, INV, true>(buf, tw, load);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 4, 64, INV, false>(buf, tw, none);
    }
}

// Same schedule with the data already in `buf` (the caller staged it and synchronised).
template <int N, bool INV>
__device__ __forceinline__ void fft_block_inplace(float2 *buf, const float2 *tw) {
    auto none = [](int) { return make_float2(0.f, 0.f); };
    if constexpr (N == 4096) {
        fft_pass<N, 8, 1, INV, false>(buf, tw, none);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 8, 64, INV, false>(buf, tw, none);
        fft_pass<N, 8, 512, INV, false>(buf, tw, none);
    } else if constexpr (N == 2048) {
        fft_pass<N, 8, 1, INV, false>(buf, tw, none);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 8, 64, INV, false>(buf, tw, none);
        fft_pass<N, 4, 512, INV, false>(buf, tw, none);
    } else {
        static_assert(N == 1024, "unsupported FFT size");
        fft_pass<N, 8, 1, INV, false>(buf, tw, none);
        fft_pass<N, 8, 8, INV, false>(buf, tw, none);
        fft_pass<N, 4, 64, INV, false>(buf, tw, none);
        fft_pass<N, 4, 256, INV, false>(buf, tw, none);
    }
}
