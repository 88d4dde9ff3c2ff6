// Internal interface of the FFT-domain convolution form (amt_fftconv.hip), used by amt_rdcnn.hip (conv mode 3).
#pragma once
#include "amt_common.h"

struct amt_fftconv_layer;

struct FcEpilogue {
    const float *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;   // device [32]
    const float *sc = nullptr; size_t sc_stride = 0;                          // identity shortcut [B][H][W][32]
    const float *sc1 = nullptr; size_t sc1_stride = 0;                        // rank-1 shortcut input [B][H][W]
    const float *sc1_w = nullptr, *sc1_s = nullptr, *sc1_t = nullptr;         // device [32]
};

// kernel: host [4][16][32][32] (Keras layout kh, kw, cin, cout)
int amt_fftconv_layer_create_internal(amt_fftconv_layer **out, const float *kernel);
void amt_fftconv_layer_destroy_internal(amt_fftconv_layer *L);
size_t amt_fftconv_freq_floats(int B, int H);                                // floats of one [289][B][H][64] tensor
int amt_fftconv_forward_fft(const amt_fftconv_layer *L, const float *in_sp, size_t in_stride, int B, int H, int W,
                            float *Xf, float *amaxf, hipStream_t st);
int amt_fftconv_gemm(const amt_fftconv_layer *L, const float *Xf, const float *amaxf, int B, int H, float *Yf, hipStream_t st);
int amt_fftconv_inverse_epilogue(const amt_fftconv_layer *L, const float *Yf, const FcEpilogue &ep, int B, int H, int W,
                                 float *out_sp, size_t out_stride, float *Xf_next, float *amaxf_next, float *amax_out,
                                 hipStream_t st);

// ---- packed-image form of the 10 x 64, 64 -> 64 layers (amt_fftpk.hip) ----------------------------------------------
struct amt_fftpk_layer;
// kernel: host [4][16][64][64] (Keras layout kh, kw, cin, cout); synchronises the null stream (weights are transformed on the device)
int amt_fftpk_layer_create_internal(amt_fftpk_layer **out, const float *kernel);
void amt_fftpk_layer_destroy_internal(amt_fftpk_layer *L);
size_t amt_fftpk_freq_floats(int B);                                         // floats of one [B][577][128] tensor
int amt_fftpk_forward_fft(const amt_fftpk_layer *L, const float *in_sp, size_t in_stride, int B, float *Xf, float *amaxf, hipStream_t st);
int amt_fftpk_gemm(const amt_fftpk_layer *L, const float *Xf, const float *amaxf, int B, float *Yf, hipStream_t st);
int amt_fftpk_inverse_epilogue(const amt_fftpk_layer *L, const float *Yf, const FcEpilogue &ep, int B, float *out_sp, size_t out_stride,
                               float *Xf_next, float *amaxf_next, float *amax_out, hipStream_t st);
