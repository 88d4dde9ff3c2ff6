/*
 * amt_saga.h -- C ABI of the MI355X (gfx950) hot path of AMT-SAGA.
 *
 * Drop-in boundary.  The reference (RobertKajnak/AMT-SAGA) is pure Python and
 * has no FFI; the boundary it does have is the object surface that
 * training.py calls.  Each entry point below names the
 * reference interface it replaces (paths relative to the upstream repo).  The
 * Python host shim in amt-saga_amd/amt_saga binds these with ctypes and keeps
 * the reference's class / method names on top.
 *
 * Conventions
 *   - every function returns an int status: AMT_OK (0) or a negative AMT_E_*;
 *     no exceptions cross the ABI.  amt_strerror() gives the text the Python
 *     shim raises as ValueError / RuntimeError (the reference raises
 *     ValueError, util_train_test.py:109-112, util_audio.py:207,505).
 *   - all data pointers are DEVICE pointers owned by the caller; sizes are
 *     explicit; `stream` is a hipStream_t passed as void*.  Calls enqueue work
 *     and return; nothing synchronises the device except amt_*_create.
 *   - spectra are FRAME-MAJOR in HBM:  spec[b][t][f], f contiguous, row pitch
 *     `ldf` floats (ldf >= n_fft/2+1, multiple of 4), window pitch
 *     `spec_stride` elements.  The reference's numpy layout is [f][t]; the
 *     Python shim transposes at the boundary.  Pad columns f in [F, ldf) are
 *     written as zero by every producer.
 *   - no hidden global state besides the handles returned by *_create.
 */
#ifndef AMT_SAGA_H
#define AMT_SAGA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMT_OK            0
#define AMT_E_INVALID    -1   /* bad argument (null pointer, non power-of-two n_fft, ...) */
#define AMT_E_SHAPE      -2   /* shape mismatch ("Invalid Input shape", util_train_test.py:109) */
#define AMT_E_HIP        -3   /* a HIP runtime call failed; see amt_last_hip_error() */
#define AMT_E_NOMEM      -4
#define AMT_E_UNSUPPORTED -5  /* topology / size outside what the kernels are built for */
#define AMT_E_ATTRIB     -6   /* "Requested attribute does not exist" (util_audio.py:207) */

int         amt_version(void);                 /* ABI version, currently 1 */
const char *amt_strerror(int status);
const char *amt_last_hip_error(void);          /* text of the last failing HIP call (thread-local) */
int         amt_device_info(int *cu_count, int *lds_bytes, char *arch, int arch_len);

/* ------------------------------------------------------------------------ *
 * STFT / iSTFT   (replaces librosa.stft / magphase / istft as called from
 * audio_complete.F / .mag / .ph / .wf, util_audio.py:88-168)
 * ------------------------------------------------------------------------ */
typedef struct amt_stft_plan amt_stft_plan;

/* n_fft: power of two in [256, 4096]; hop > 0; center as librosa (reflect pad
 * n_fft/2).  Uploads the twiddle / window table (computed in double). */
int amt_stft_plan_create(amt_stft_plan **plan, int n_fft, int hop, int center);
int amt_stft_plan_destroy(amt_stft_plan *plan);
/* frames produced for a signal of L samples: center ? 1 + L/hop : 1 + (L-n_fft)/hop */
int amt_stft_frames(const amt_stft_plan *plan, int L);

/* audio_complete.mag / .ph / .ref_mag (util_audio.py:139-174) in one pass:
 *   wave  [B][wave_stride]  f32, L valid samples per window
 *   mag   [B][T][ldf]       f32 = |STFT|
 *   phase [B][T][ldf]       float2 = exp(i*angle(STFT)) (1+0i where STFT == 0); may be NULL
 *   ref_max [B]             f32 = max(mag) per window (np.max(self.mag), :173); may be NULL
 * T must equal amt_stft_frames(plan, L). */
int amt_stft_mag(const amt_stft_plan *plan, const float *wave, int B, int L,
                 size_t wave_stride, float *mag, float *phase_ri, float *ref_max,
                 int T, int ldf, size_t spec_stride, void *stream);

/* audio_complete.wf getter, the `mag * ph -> librosa.istft` branch
 * (util_audio.py:94-97): wave_out[b][0 .. hop*(T-1)) f32.  phase_ri may be
 * NULL (then `mag` is taken as interleaved complex F with pitch 2*ldf). */
int amt_istft(const amt_stft_plan *plan, const float *mag, const float *phase_ri,
              int B, int T, int ldf, size_t spec_stride, float *wave_out,
              size_t wave_stride, void *stream);

/* np.max over each window's [T][ldf] block (audio_complete.ref_mag, :170-174) */
int amt_window_max(const float *spec, int B, int T, int ldf, size_t spec_stride,
                   float *out_max, void *stream);

/* ------------------------------------------------------------------------ *
 * Spectral subtraction  (replaces audio_complete.subtract, util_audio.py:221-259)
 * ------------------------------------------------------------------------ */
typedef struct amt_subtract_args {
    float       *resid;          /* [B][T][ldf] in place (self.mag)                          */
    const float *resid_max;      /* [B] current ref_mag of each window, or NULL              */
    const float *guess;          /* base of the guess magnitudes, frame-major [.][Tg][ldf]   */
    const float *guess_max;      /* [n_guess] ref_mag of each guess, or NULL                 */
    const int32_t *guess_index;  /* [B] which guess each window subtracts; NULL => b         */
    const int32_t *guess_frames; /* [B] frames of each window's guess; NULL => guess_frames_all */
    const int32_t *offset_frames;/* [B] first frame (already max(..-attack_compensation,0));
                                    NULL => 0                                                */
    float       *new_max;        /* [B] out: np.max(self.mag) after the step, or NULL        */
    size_t       resid_stride;   /* elements between windows                                 */
    size_t       guess_stride;   /* elements between guesses                                 */
    int32_t      B, T, ldf, F;
    int32_t      guess_frames_all;
    int32_t      normalize;      /* scale by resid_max/guess_max (needs both arrays)         */
    int32_t      relu;
    float        overkill_factor;
} amt_subtract_args;

int amt_subtract(const amt_subtract_args *args, void *stream);

/* The same step on the frames that change (util_audio.py:250-259 writes only self.mag[:, off:off + Tg]): reads and
 * writes frames [offset, offset + guess_frames) of every window, updates their entries of frame_max [B][T] and sets
 * new_max[b] = max over t of frame_max[b][t].  Preconditions, the caller's: relu != 0; resid >= 0 everywhere (magnitudes,
 * or the output of an earlier step), so that the ReLU leaves the other frames as they are; frame_max[b][t] = the maximum
 * of resid's frame t over the F bins (amt_compress_bands_fmax leaves it).  span_cap >= every window's guess frame count
 * (the guess tensor's frames).  Residual and maxima are bit-identical to amt_subtract's. */
int amt_subtract_span(const amt_subtract_args *args, float *frame_max, int span_cap, void *stream);

/* ------------------------------------------------------------------------ *
 * Feature gathers (audio_complete.compress_bands :436-466, ._resize :384-409,
 * .resize :469-507, .section_power :334-349 and the recipe of
 * training.py:333-363).  Outputs are the NHWC head inputs [B][bands][frames].
 * ------------------------------------------------------------------------ */
/* C_timing = _resize(compress_bands(mag, bands), target) / ref   (training.py:333-336)
 *   edges [bands+1] int32 device (row ranges over f: util_audio.py:451-456)
 *   ref [B] device or NULL (=> 1): the song-level ref_mag the recipe divides by
 *   src_frame [target_frames] device or NULL (=> identity): source frame of every
 *   output column (-1 => zero column), i.e. _resize's tile/crop rule as a table
 *   out [B][bands][target_frames] f32 */
int amt_compress_bands(const float *mag, int B, int T, int F, int ldf, size_t spec_stride,
                       const int32_t *edges, int bands, const float *ref,
                       const int32_t *src_frame, float *out, int target_frames,
                       void *stream);
/* The same, also leaving frame_max [B][T] = max over the F bins of every frame (the kernel has the frame in registers):
 * what amt_subtract_span needs.  Identity frame map only (src_frame NULL, target_frames == T). */
int amt_compress_bands_fmax(const float *mag, int B, int T, int F, int ldf, size_t spec_stride,
                            const int32_t *edges, int bands, const float *ref,
                            const int32_t *src_frame, float *out, int target_frames,
                            float *frame_max, void *stream);

/* short-window gather: for every window b take frames src_frame[b][0..frames)
 * (-1 => zeros; the table realises _resize's tile/crop rule) and rows
 * [band_min[b], band_min[b]+bands) (zero past F; band_min NULL => 0):
 *   mode 0: out = mag / ref[b]  (ref NULL => 1)             (F_sw_inst_foc / _const, :350,:360)
 *   mode 1: out = log10(mag*1000+1) / max over the window  (.._log10, :353-354,:358-359)
 *   mode 2: out = (angle(phase)+3.15)/6.3                  (ph, :361-362)
 * out [B][bands][frames] f32. */
int amt_short_window(const float *mag, const float *phase_ri, int B, int T, int F,
                     int ldf, size_t spec_stride, const int32_t *src_frame,
                     int frames, const int32_t *band_min, int bands,
                     const float *ref, int mode, float *out, void *stream);

/* window management as index maps: audio_complete.section :286-328, .slice :351-365, .concat :374-382,
 * .resize :469-507 (column selection with zero padding) and .section_power :334-349 (band of bins):
 *   out[b][j][k] = src[b][src_frame[b*table_stride + j]][band_min + k]
 * zero where the frame index is < 0 or >= T, or the bin is >= F or k >= bands.  elem = floats per bin
 * (1 = magnitude / dB, 2 = unit phase or complex F).  table_stride 0 = one table for all windows.
 * out rows have ldf_out bins (>= bands); the pad is written as zero. */
int amt_gather_frames(const float *src, int B, int T, int F, int ldf_src, size_t src_stride, int elem,
                      const int32_t *src_frame, int table_stride, int n_out, int band_min, int bands,
                      float *out, int ldf_out, size_t out_stride, void *stream);

/* audio_complete.D getter (util_audio.py:176-180 -> librosa.amplitude_to_db(mag, ref=ref_mag)):
 *   D = 20 log10(max(amin, |mag|)) - 20 log10(max(amin, ref[b])), floored at max(D) - top_db
 * window_max[b] = max(mag[b]) (amt_window_max or the fused STFT max) supplies max(D); top_db < 0 = no floor.
 * librosa defaults: amin = 1e-5, top_db = 80. */
int amt_amplitude_to_db(const float *mag, int B, int T, int F, int ldf, size_t spec_stride, const float *ref,
                        const float *window_max, float amin, float top_db, float *out_db, void *stream);
/* librosa.db_to_amplitude as used by the mag / F / wf getters when only D is set (util_audio.py:101,124,145):
 *   mag = ref[b] * 10^(D / 20) */
int amt_db_to_amplitude(const float *db, int B, int T, int F, int ldf, size_t spec_stride, const float *ref,
                        float *out_mag, void *stream);

/* audio_complete.spectral_flatness (util_audio.py:330-332 -> librosa.feature.spectral_flatness, power 2):
 * out[b][t] = exp(mean_f log max(amin, mag^2)) / mean_f max(amin, mag^2); the reference takes np.mean over t
 * and skips a song whose render is white noise (> 0.3, training.py:266).  librosa default amin = 1e-10. */
int amt_spectral_flatness(const float *mag, int B, int T, int F, int ldf, size_t spec_stride, float amin,
                          float *out, void *stream);

/* ------------------------------------------------------------------------ *
 * Constant-Q slices (replaces audio_complete.slice_C, util_audio.py:411-434,
 * i.e. |librosa.cqt| evaluated only at the frames _resize keeps).  The
 * transform is the build's own spec (oracle/cqt.py); parity vs librosa is
 * unpinned.
 * ------------------------------------------------------------------------ */
typedef struct amt_cqt_args {
    const float   *wave;        /* [B][wave_stride], L valid samples                    */
    const int32_t *src_frame;   /* [B][frames] STFT-frame index of each output column, -1 => zeros */
    const int32_t *bin0;        /* [B] first bin of the table for each window, or NULL => 0 */
    const uint32_t *phase_inc;  /* [n_table] per-bin frequency, cycles/sample * 2^32    */
    const int32_t *length;      /* [n_table] filter length N_k in samples               */
    const float   *ref;         /* [B] divisor (ref_C_*, training.py:340-388) or NULL    */
    const float   *coef;        /* [n_table][192] per-bin phasor table from amt_cqt_coef() */
    float         *out;         /* [B][n_bins][frames]                                   */
    size_t         wave_stride;
    int32_t        B, L, hop, frames, n_bins, n_table;
} amt_cqt_args;

/* Block-sum kernel (needs coef) when hop is a power of two in 128..2048 and L/hop <= ~1300 (the block sums live
 * in LDS); otherwise every frame is summed directly over its own N_k samples (any L and hop, coef unused). */
int amt_cqt_slices(const amt_cqt_args *args, void *stream);

/* Complex form (util_audio.py:424-429 with magnitude_only=False: librosa.cqt's complex output is returned as is):
 * args->out receives the real parts, out_im the imaginary parts, both [B][n_bins][frames].  The phase refers to the
 * frame's centre (sample t * hop), where the analysis filter's own phase is zero -- the convention of a centred filter
 * bank; like the magnitudes, the values follow the build's CQT definition (oracle/cqt.py), not librosa's recursion. */
int amt_cqt_slices_complex(const amt_cqt_args *args, float *out_im, void *stream);

/* Per-bin phasor table of a CQT grid (32 x 3 unit complex numbers per bin, f64-evaluated): computed once per
 * (phase_inc, length) table and passed to amt_cqt_slices / amt_cqt_window_max.  coef: [n_table][192] floats. */
int amt_cqt_coef(const uint32_t *phase_inc, const int32_t *length, int n_table, float *coef, void *stream);

/* Song-level CQT normalisers (training.py:271-282: ref_C_* = np.max(mid_wf.slice_C(0, duration,
 * n_frames, pitch_frames, bins_per_tone=...)), the maximum of the whole CQT): out_max[b] = max over the
 * n_bins rows of the table and over EVERY frame t = 0 .. L/hop of |C[k, t]| of signal b -- a window, or a whole
 * song.  O(L) per bin whatever the filter length (prefix sums of per-hop block sums); hop a power of two in
 * 128..2048.  Up to ~1300 hops the block sums live in LDS; longer signals need
 * amt_cqt_window_max_workspace(L, hop, n_bins, B) bytes of device scratch (AMT_E_NOMEM if it is missing). */
size_t amt_cqt_window_max_workspace(int L, int hop, int n_bins, int B);
int amt_cqt_window_max(const float *wave, int B, int L, size_t wave_stride, int hop,
                       const uint32_t *phase_inc, const int32_t *length, const float *coef, int n_bins,
                       float *out_max, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ *
 * Loop glue: predicted note -> integer decisions, gather tables, guess pick,
 * event records.  Restates per window what training.py:296-449 does with the
 * gold note (the reference never rounds: predict returns floats,
 * RDCNN.py:591-597; the rounding rules are rint / first argmax, as numpy).
 * ------------------------------------------------------------------------ */
/* out[i] = clamp(rint(x[i*stride]), lo, hi)  (half-to-even; NaN -> lo) */
int amt_round_clamp(const float *x, int n, int stride, int lo, int hi, int32_t *out,
                    void *stream);
/* out[i] = first argmax of p[i][0..K)  (np.argmax) */
int amt_argmax_rows(const float *p, int n, int K, int32_t *out, void *stream);
/* out[b][j] = source frame of column j of _resize(X[:, start[b]:end[b]], frames)
 * (util_audio.py:384-409 with numpy slice clamping to [0,T]); -1 = zero column */
int amt_resize_table(const int32_t *start, const int32_t *end, int n, int T, int frames,
                     int32_t *out, void *stream);
/* guess_index = prog_group[program]*n_pitch + (pitch-pitch_lo) (clamped);
 * guess_frames = min(max(end-onset,0) + tail_frames, bank_frames)
 * (the rendered guess = note + 1 s release, util_audio.py:876).  program /
 * prog_group may be NULL (=> group 0). */
int amt_note_select(const int32_t *program, const int32_t *pitch, const int32_t *onset,
                    const int32_t *end, const int32_t *prog_group, int n_prog, int n,
                    int pitch_lo, int n_pitch, int tail_frames, int bank_frames,
                    int32_t *guess_index, int32_t *guess_frames, void *stream);
/* events[i] = {window0+i, iter, pitch, program, velocity, onset_frame, end_frame}
 * (int32 x 7); NULL inputs are recorded as -1 */
int amt_pack_events(int n, int window0, int iter, const int32_t *pitch,
                    const int32_t *program, const int32_t *velocity, const int32_t *onset,
                    const int32_t *end, int32_t *events, void *stream);
/* out[i] = x[i]*mul + add (bin offsets of the focused CQT grids) */
int amt_affine_i32(const int32_t *x, int n, int mul, int add, int32_t *out, void *stream);

/* ------------------------------------------------------------------------ *
 * Guess synthesis (stand-in for note_sequence.render(), util_audio.py:758-786:
 * fluidsynth + soundfont are not available).  Additive synth defined in
 * amt_saga/synth.py; window scaling follows render() (:778-781).
 * ------------------------------------------------------------------------ */
/* notes [B][max_notes][5] f32 = {preset group 0..2, midi pitch (<0 = unused slot),
 * velocity, onset s, duration s}; wave [B][wave_stride] f32 out (L samples);
 * peak_scratch [B] f32 device scratch. */
int amt_synth_windows(const float *notes, int max_notes, int B, int L, float sample_rate,
                      float *wave, size_t wave_stride, float *peak_scratch, void *stream);
/* Per-program timbres ("per-program timbres"; util_audio.py:758-786 renders every MIDI program through
 * its own soundfont preset): notes[..][0] indexes a caller-supplied table timbres [n_timbres][5] f32 (device) =
 * {harmonics H, spectral slope, decay tau s (0 = sustained), attack s, weight of the even harmonics}; the build's
 * table for the 128 General MIDI programs is amt_saga/synth.py:gm_timbre_table.  timbres == NULL: the three built-in
 * groups of amt_synth_windows. */
int amt_synth_windows_timbres(const float *notes, int max_notes, int B, int L, float sample_rate,
                              const float *timbres, int n_timbres, float *wave, size_t wave_stride,
                              float *peak_scratch, void *stream);
/* SoundFont 2 sample playback ("optionally SF2 sample playback if a soundfont is supplied";
 * util_audio.py:758-786 renders through fluidsynth + main.py's -soundfont_path): notes[..][0] is the MIDI program;
 * samples [n_samples] f32 = the font's 16-bit pool / 32768; zones [n_zones][20] f32 and first [n_prog + 1] i32 as
 * amt_saga/sf2.py:SoundFont.tables() lays them out (flattened preset x instrument zones of bank 0: ranges, sample
 * and loop points, tuning, attenuation, volume envelope).  Same amplitude law and 1 s tail as amt_synth_windows.
 * The zone table is trusted (the host parser validates offsets against the pool). */
int amt_sf2_synth_windows(const float *notes, int max_notes, int B, int L, float sample_rate, const float *samples,
                          int n_samples, const float *zones, int n_zones, const int32_t *first, int n_prog,
                          float *wave, size_t wave_stride, float *peak_scratch, void *stream);
/* one guess note per window from the loop's integer decisions:
 * {prog_group[program], pitch, velocity (or default), 0, min((end-onset)*frame_seconds, max_dur)}
 * (training.py:421-424 builds the guessed note the same way, from the gold values) */
int amt_guess_notes(const int32_t *program, const int32_t *pitch, const int32_t *velocity,
                    const int32_t *onset, const int32_t *end, const int32_t *prog_group,
                    int n_prog, int n, float frame_seconds, float max_dur,
                    float default_velocity, float *notes, void *stream);

/* ------------------------------------------------------------------------ *
 * RDCNN forward (replaces res_net.predict, RDCNN.py:591-597, for the graph
 * built by RDCNN.py:176-233).
 * ------------------------------------------------------------------------ */
typedef struct amt_rdcnn amt_rdcnn;

typedef struct amt_rdcnn_desc {
    int32_t n_towers;                    /* 1 or 2 (instrument_dual)                    */
    int32_t in_h[2], in_w[2];            /* input_shapes (bands, frames), Cin = 1        */
    int32_t kh[2], kw[2];                /* kernel_sizes                                 */
    int32_t pool_h[2], pool_w[2];        /* pool_sizes                                   */
    int32_t conv_layers;                 /* convolutional_layer_count                    */
    int32_t feature_expand_frequency;
    int32_t pool_layer_frequency;
    int32_t residual_frequency;          /* residual_layer_frequencies = [r]; 0 = none   */
    int32_t dense_units;                 /* 300                                          */
    int32_t output_classes;              /* 1 => sigmoid + range scaling; >1 => softmax  */
    float   out_lo, out_hi;              /* output_range                                 */
} amt_rdcnn_desc;

/* Weights: one host blob of f32 in the canonical order documented in
 * amt-saga_amd/amt_saga/rdcnn.py (pack_weights); n_floats is checked against
 * the descriptor.  Uploads and pre-arranges them for the MFMA kernels. */
int amt_rdcnn_create(amt_rdcnn **net, const amt_rdcnn_desc *desc,
                     const float *weights_host, size_t n_floats);
int amt_rdcnn_destroy(amt_rdcnn *net);
size_t amt_rdcnn_param_count(const amt_rdcnn_desc *desc);
/* device scratch needed for a batch of B windows */
size_t amt_rdcnn_workspace_bytes(const amt_rdcnn *net, int B);
/* x[t]: [B][in_h][in_w] f32 (NHWC with C = 1) per tower; y: [B][output_classes] f32
 * (scaled regression value or softmax probabilities); logits (optional, may be
 * NULL): [B][output_classes] pre-activation of the last Dense. */
int amt_rdcnn_forward(const amt_rdcnn *net, const float *const *x, int B,
                      float *y, float *logits, void *workspace,
                      size_t workspace_bytes, void *stream);
/* Arithmetic of the convolutions: 0 = v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate);
 * 1 = "split-bf16": every f32 operand is split exactly into three bf16 terms and the
 * six significant cross products run on v_mfma_f32_32x32x16_bf16 with f32
 * accumulation -- same accuracy class as f32 (dropped terms <= 2^-26), 2.67x fewer
 * matrix-pipe cycles.  Layers the split kernel is not built for keep mode 0. */
int amt_rdcnn_set_mode(amt_rdcnn *net, int mode);
/* FLOPs (2*MAC) of one window's forward, for roofline accounting */
double amt_rdcnn_flops_per_window(const amt_rdcnn *net);
/* Measurement hook: when enabled, every convolution launch of amt_rdcnn_forward
 * is bracketed by HIP events on the caller's stream.  amt_rdcnn_profile_read
 * waits for the recorded events and returns one row per conv layer:
 * desc[r] = {tower, layer, kh, kw, cin, cout, H, W}, ms[r] = summed kernel time,
 * windows[r] = windows processed, flops_per_window[r] = 2*H*W*kh*kw*cin*cout.
 * n_rows receives the number of layers (call with cap = 0 to size buffers). */
int amt_rdcnn_profile(amt_rdcnn *net, int enable);
int amt_rdcnn_profile_read(amt_rdcnn *net, int32_t *desc, double *ms, double *windows,
                           double *flops_per_window, int cap, int *n_rows, int reset);

/* ------------------------------------------------------------------------ *
 * RDCNN training step (replaces res_net.train / res_net.test, RDCNN.py:503-526, :559-589: Keras
 * train_on_batch / test_on_batch of the model compiled at RDCNN.py:245-254 with Adagrad and
 * mean_squared_error (one output) or sparse_categorical_crossentropy).
 * ------------------------------------------------------------------------ */
typedef struct amt_trainer amt_trainer;

/* weights_host: the canonical blob of amt_rdcnn_create (trainable tensors AND the BN moving statistics).
 * lr / epsilon <= 0 select Keras' Adagrad defaults (0.01, 1e-7).  initial_accumulator: starting value of the
 * squared-gradient accumulators -- 0 in Keras 2.2 / tf.keras 1.13, 0.1 in tf.keras >= 1.14 (the reference's
 * `keras.optimizers.Adagrad()`, RDCNN.py:246-253, pins neither version). */
int amt_trainer_create(amt_trainer **trainer, const amt_rdcnn_desc *desc, const float *weights_host,
                       size_t n_floats, float lr, float epsilon, float initial_accumulator);
int amt_trainer_destroy(amt_trainer *trainer);
/* One batch.  x[t]: device [B][in_h][in_w] per tower; y: device [B] f32 -- the class index (softmax heads)
 * or the target ALREADY scaled to the activation range (RDCNN.py:304-306, :513-514).
 * update != 0: train_on_batch -- BatchNormalization with batch statistics (moving statistics updated with
 *   momentum 0.99), backward pass, Adagrad step.
 * update == 0: test_on_batch -- inference-mode forward and the loss only.
 * loss_host (may be NULL): the batch loss; reading it synchronises the stream.
 * pred (may be NULL): device [B][output_classes], sigmoid activation / softmax probabilities. */
int amt_trainer_step(amt_trainer *trainer, const float *const *x, const float *y, int B, int update,
                     float *loss_host, float *pred, void *stream);
/* canonical blob back to the host (synchronises): current weights / the gradients of the last step
 * (zeros for the BN moving statistics) */
int amt_trainer_get_weights(amt_trainer *trainer, float *weights_host, size_t n_floats);
int amt_trainer_get_grads(amt_trainer *trainer, float *grads_host, size_t n_floats);

/* MFMA form of amt_cqt_window_max for batches of windows (np.max(mid_wf.slice_C(0, duration, n_frames, ...)),
 * training.py:271-282, with each window standing for its song): the per-block sums as one split-fp16 GEMM per window
 * against a per-bin phasor table on a bin-independent block grid (amt_cqt.hip).  Same definition, same result to the
 * 1e-4 the VALU form is held to; hop a power of two in 256..2048 and at most 544 hop-blocks per window
 * (AMT_E_UNSUPPORTED otherwise: use amt_cqt_window_max).
 *   amt_cqt_mfma_table_bytes(hop, n_bins)   bytes of the device phasor table for a bin grid
 *   amt_cqt_mfma_table(...)                 builds it (once per grid and hop)
 *   amt_cqt_window_max_mfma(...)            out_max[b] = max over bins and frames 0 .. L / hop; amax_scratch: [B] f32 */
size_t amt_cqt_mfma_table_bytes(int hop, int n_bins);
int amt_cqt_mfma_table(const uint32_t *phase_inc, const int32_t *length, int n_bins, int hop, void *table,
                       void *stream);
int amt_cqt_window_max_mfma(const float *wave, int B, int L, size_t wave_stride, int hop,
                            const uint32_t *phase_inc, const int32_t *length, const void *table, int n_bins,
                            float *out_max, float *amax_scratch, void *stream);

/* ------------------------------------------------------------------------ *
 * FFT-domain form of ONE Conv2D(32 -> 32, (4, 16), 'same') + BatchNormalization + sigmoid (+ Add + BatchNormalization)
 * layer of res_net (RDCNN.py:186-193) on an H x W image, W <= 561: the stand-alone entry the parity tests and the
 * layer microbenchmark use; inside amt_rdcnn_forward the same kernels run chained (conv mode 3).
 *   kernel_host [4][16][32][32] (kh, kw, cin, cout); s1, t1: folded BN (+ bias) [32]; s2, t2: the BN after the Add, or
 *   NULL; in / shortcut / out: device [B][H][W][32]; workspace: amt_fftconv_workspace_bytes(B, H) of device memory.
 * ------------------------------------------------------------------------ */
typedef struct amt_fftconv_layer amt_fftconv_layer;
int amt_fftconv_create(amt_fftconv_layer **layer, const float *kernel_host, const float *s1, const float *t1,
                       const float *s2, const float *t2);
int amt_fftconv_destroy(amt_fftconv_layer *layer);
size_t amt_fftconv_workspace_bytes(int B, int H);
int amt_fftconv_run(const amt_fftconv_layer *layer, const float *in, const float *shortcut, int B, int H, int W,
                    float *out, void *workspace, size_t workspace_bytes, int repeat_gemm, void *stream);

/* ------------------------------------------------------------------------ *
 * The same layer kind on the 10 x 64 images behind the first pooling of the timing classifier
 * (timing_classifier.py:13-36; Conv2D(64, (4, 16), 'same') + BatchNormalization + sigmoid (+ Add + BatchNormalization),
 * RDCNN.py:186-198) in the packed-image FFT-domain form (amt_fftpk.hip): the whole image is one 1152-point sequence per
 * channel pair, one 128 x 128 GEMM per frequency pair with the windows as rows.  Stand-alone entry for the parity tests
 * and the layer microbenchmark; inside amt_rdcnn_forward the same kernels run chained (conv mode 3).
 *   kernel_host [4][16][64][64]; s1, t1, s2, t2 [64]; in / shortcut / out: device [B][10][64][64];
 *   chain: extra applications of the layer (BN + sigmoid, no shortcut) with the hand-over in the frequency domain before the
 *   final one; repeat_gemm: the final GEMM launched that many times (timing).
 * ------------------------------------------------------------------------ */
typedef struct amt_fftpk_layer amt_fftpk_layer;
int amt_fftpk_create(amt_fftpk_layer **layer, const float *kernel_host, const float *s1, const float *t1,
                     const float *s2, const float *t2);
int amt_fftpk_destroy(amt_fftpk_layer *layer);
size_t amt_fftpk_workspace_bytes(int B);
int amt_fftpk_run(const amt_fftpk_layer *layer, const float *in, const float *shortcut, int B, float *out,
                  void *workspace, size_t workspace_bytes, int chain, int repeat_gemm, void *stream);

/* ------------------------------------------------------------------------ *
 * Measurement probe (no reference counterpart; bench.py's roofline object).  Sustained rate of back-to-back
 * v_mfma_f32_16x16x32_f16 on register operands with `waves_per_simd` waves per SIMD on every CU, `iters` x 8
 * MFMAs per wave and launch, best of `launches`; random_operands != 0: random non-zero f16 operands (the rate
 * the power limit allows on live data), 0: zeros (the cycle-limited rate).  Synchronises the stream.
 * ------------------------------------------------------------------------ */
int amt_probe_mfma_f16(int waves_per_simd, int iters, int random_operands, int launches,
                       double *tflops_best, double *ms_best, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AMT_SAGA_H */
