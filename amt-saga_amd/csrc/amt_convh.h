// Internal interface of the split-fp16 convolution kernels in their trainer form (conv_f16x3s_kernel<.., TRAIN = true>,
// amt_conv_f16x3.h; launchers in amt_rdcnn.hip), used by amt_train.hip for the forward convolution and the data
// gradient of a training step.  Everything here takes device pointers; nothing synchronises.
#pragma once
#include "amt_common.h"

struct amt_convh_plan {
    int kh = 0, kw = 0, cin = 0, cout = 0, H = 0, W = 0;      // the convolution that RUNS (a data gradient: cin = the layer's Cout)
    int TH = 0, TW = 0, NWIN = 1, masked = 0;
    size_t lds = 0;
};
// AMT_OK, or AMT_E_UNSUPPORTED when no instantiation covers the shape (the caller keeps its own path)
int amt_convh_plan_init(amt_convh_plan *pl, int kh, int kw, int cin, int cout, int H, int W);
size_t amt_convh_packed_bytes(const amt_convh_plan *pl);

// One weight-preparation job per layer: w is the layer's f32 kernel [kh kw][Cin][Cout] (Keras layout);
// packed_fwd is the fragment layout of the layer's own convolution (cin = Cin, cout = Cout), packed_bwd (may be null)
// that of its data gradient (kernel flipped in both directions, channels transposed: cin = Cout, cout = Cin).
// wmax / sw: device scalars of the layer (max |w|, its exponent 2^sw with max |w| 2^sw in [8, 16)).
struct amt_convh_pack_job {
    const float *w;
    void *packed_fwd, *packed_bwd;
    float *wmax;
    int *sw;
    int ntap, Cin, Cout;
};
// jobs: DEVICE array of n jobs; wmax slots must be zero on entry.  Two launches for all layers.
int amt_convh_pack_all(const amt_convh_pack_job *jobs_dev, int n, int max_ntap_cin_cout, hipStream_t st);

// out[b][h][w][cout] = sum over taps and cin of in(b, h + dy - pad_t, w + dx - pad_l, cin) * W(dy, dx, cin, cout)  (+ bias[cout])
// (+ acc[b][h][w][cout] when acc != null; acc may alias out).  amax_in: device [B], max |in| per window (amt_convh_absmax).
int amt_convh_run(const amt_convh_plan *pl, const float *in, float *out, const float *acc, int B, const void *packed,
                  const int *sw_dev, const float *bias, const float *amax_in, int pad_t, int pad_l, hipStream_t st);
// amax[b] = max |x[b][0 .. n)|; amax must be zero on entry
int amt_convh_absmax(const float *x, size_t n, int B, float *amax, hipStream_t st);
