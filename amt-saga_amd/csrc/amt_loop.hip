// Small integer/index kernels that keep the detect -> subtract loop on the device:
// head outputs (floats) -> note decisions (ints), _resize gather tables, guess
// selection, event records.  The reference has no predict loop (main.py only
// trains); these restate, per window, what training.py:296-449 does with the
// gold note, using the predicted one.
#include "amt_common.h"

// y = clamp(rint(x), lo, hi): numpy rint (half to even) == rintf
__global__ void round_clamp_kernel(const float *__restrict__ x, int n, int stride, int lo, int hi,
                                   int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = rintf(x[(size_t)i * stride]);
    int r = v < (float)lo ? lo : (v > (float)hi ? hi : (int)v);
    if (!(v == v)) r = lo;                               // NaN -> lo
    out[i] = r;
}

// argmax over K classes, first maximum (numpy argmax)
__global__ void argmax_rows_kernel(const float *__restrict__ p, int n, int K,
                                   int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = p + (size_t)i * K;
    int best = 0;
    float bv = r[0];
    for (int k = 1; k < K; ++k)
        if (r[k] > bv) { bv = r[k]; best = k; }
    out[i] = best;
}

// source frame of every column of _resize(X[:, s:t], frames) (util_audio.py:384-409,
// :431-434); -1 = zero column.  One thread per (window, column).
__global__ void resize_table_kernel(const int32_t *__restrict__ s_, const int32_t *__restrict__ t_,
                                    int n, int T, int frames, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * frames) return;
    const int b = i / frames, j = i - b * frames;
    int s = s_[b], t = t_[b];
    s = s < 0 ? 0 : (s > T ? T : s);                     // numpy slice clamping of C[:, s:t]
    t = t < s ? s : (t > T ? T : t);
    const int len = t - s;
    int src;
    if (len == 0) src = -1;
    else if (len == frames || len > frames) src = s + j;
    else if (len < 3) src = j == 0 ? s : s + len - 1;
    else {
        const int inner = len - 2;                       // lim == 1 (util_audio.py:395)
        const int reps = (frames - 2) / inner;
        const int mid = inner * reps;
        const int n_tail = frames - mid - 1;
        if (j == 0) src = s;
        else if (j <= mid) src = s + 1 + (j - 1) % inner;
        else src = s + len - n_tail + (j - 1 - mid);
    }
    out[i] = src;
}

// guess-bank row and length for the predicted note; event record
//   guess row   = prog_group[program] * n_pitch + (pitch - pitch_lo)
//   guess frames = min(max(end - onset, 0) + tail_frames, bank_frames)
//   vel_bin0    = bins_per_semitone * (pitch - pitch_lo)  (first bin of the velocity CQT grid)
__global__ void note_select_kernel(const int32_t *__restrict__ program, const int32_t *__restrict__ pitch,
                                   const int32_t *__restrict__ onset, const int32_t *__restrict__ end,
                                   const int32_t *__restrict__ prog_group, int n_prog, int n, int pitch_lo,
                                   int n_pitch, int tail_frames, int bank_frames,
                                   int32_t *__restrict__ guess_index, int32_t *__restrict__ guess_frames) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int pr = program ? program[i] : 0;
    pr = pr < 0 ? 0 : (pr >= n_prog ? n_prog - 1 : pr);
    const int g = prog_group ? prog_group[pr] : 0;
    int pi = pitch[i] - pitch_lo;
    pi = pi < 0 ? 0 : (pi >= n_pitch ? n_pitch - 1 : pi);
    guess_index[i] = g * n_pitch + pi;
    int d = end[i] - onset[i];
    d = d < 0 ? 0 : d;
    d += tail_frames;
    guess_frames[i] = d > bank_frames ? bank_frames : d;
}

__global__ void pack_events_kernel(int n, int window0, int iter, const int32_t *__restrict__ pitch,
                                   const int32_t *__restrict__ program, const int32_t *__restrict__ velocity,
                                   const int32_t *__restrict__ onset, const int32_t *__restrict__ end,
                                   int32_t *__restrict__ events /* [n][7] */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t *e = events + (size_t)i * 7;
    e[0] = window0 + i;
    e[1] = iter;
    e[2] = pitch ? pitch[i] : -1;
    e[3] = program ? program[i] : -1;
    e[4] = velocity ? velocity[i] : -1;
    e[5] = onset ? onset[i] : -1;
    e[6] = end ? end[i] : -1;
}

__global__ void affine_i32_kernel(const int32_t *__restrict__ x, int n, int mul, int add,
                                  int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * mul + add;
}

extern "C" {

int amt_round_clamp(const float *x, int n, int stride, int lo, int hi, int32_t *out, void *stream) {
    if (!x || !out || n <= 0 || stride <= 0 || lo > hi) return AMT_E_INVALID;
    round_clamp_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, n, stride, lo, hi, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_argmax_rows(const float *p, int n, int K, int32_t *out, void *stream) {
    if (!p || !out || n <= 0 || K <= 0) return AMT_E_INVALID;
    argmax_rows_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(p, n, K, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_resize_table(const int32_t *start, const int32_t *end, int n, int T, int frames,
                     int32_t *out, void *stream) {
    if (!start || !end || !out || n <= 0 || T <= 0 || frames < 3) return AMT_E_INVALID;
    resize_table_kernel<<<(n * frames + 255) / 256, 256, 0, (hipStream_t)stream>>>(start, end, n, T, frames, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_note_select(const int32_t *program, const int32_t *pitch, const int32_t *onset,
                    const int32_t *end, const int32_t *prog_group, int n_prog, int n, int pitch_lo,
                    int n_pitch, int tail_frames, int bank_frames, int32_t *guess_index,
                    int32_t *guess_frames, void *stream) {
    if (!pitch || !onset || !end || !guess_index || !guess_frames || n <= 0 || n_pitch <= 0 || n_prog <= 0)
        return AMT_E_INVALID;
    note_select_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        program, pitch, onset, end, prog_group, n_prog, n, pitch_lo, n_pitch, tail_frames, bank_frames,
        guess_index, guess_frames);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_pack_events(int n, int window0, int iter, const int32_t *pitch, const int32_t *program,
                    const int32_t *velocity, const int32_t *onset, const int32_t *end,
                    int32_t *events, void *stream) {
    if (!events || n <= 0) return AMT_E_INVALID;
    pack_events_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, window0, iter, pitch, program,
                                                                          velocity, onset, end, events);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_affine_i32(const int32_t *x, int n, int mul, int add, int32_t *out, void *stream) {
    if (!x || !out || n <= 0) return AMT_E_INVALID;
    affine_i32_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, n, mul, add, out);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

}  // extern "C"
