// Additive note synthesiser on the GPU: the stand-in for
// note_sequence.render() -- fluidsynth + a GM soundfont, neither available
// (/root/reference/util_audio.py:758-786).  Same definition as amt_saga/synth.py:
//   note(g, p, v, t0, d)(t) = env_g(t - t0; d) * sum_h h^-slope_g sin(2 pi h f_p (t - t0)),  h f_p < sr/2
//   env = min((t-t0)/attack, 1) * exp(-(t-t0)/tau) [decaying presets] * exp(-max(t-t0-d, 0)/60 ms),
//         zero before t0 and after t0 + d + 1 s (the reference's 1 s release tail, :876)
//   window = sum_notes (v/128)^4 note;  scaled as render() does (:778-781):
//            wf * (vel_max/128)^4 / max|wf|,  vel_max - 12 (>= 1) for a single note.
// One thread per sample; oscillator phase in double (6 s x 22 kHz needs more than f32),
// sin through v_sin_f32 on the reduced phase; peak by block max + atomicMax; a second tiny
// kernel applies the scale.  Used to render the benchmark windows, the guess bank and, in
// the loop's "render" guess mode, one guess per window and iteration.
#include "amt_common.h"

// timbre = {H, slope, tau (0 = sustain), attack, even}: harmonic h weighs h^-slope, times `even` when h is even
// (reeds and stopped pipes are weak there).  The three built-in groups, or a caller's table (amt_synth_windows_timbres:
// one row per General MIDI program, amt_saga/synth.py:gm_timbre_table).
struct Preset { int H; float slope, tau, attack, even; };
__constant__ Preset c_presets[3] = {
    {12, 1.5f, 0.6f, 0.002f, 1.0f},      // piano
    {16, 1.0f, 0.0f, 0.08f, 1.0f},       // strings (tau 0 = sustain)
    {10, 1.2f, 0.35f, 0.002f, 1.0f},     // guitar
};
#define SYN_RELEASE_TAU 0.06f
#define SYN_TAIL 1.0f

__global__ __launch_bounds__(256) void synth_kernel(const float *__restrict__ notes, int max_notes,
                                                     float *__restrict__ wave, size_t wave_stride, int L,
                                                     float sr, float *__restrict__ peak,
                                                     const float *__restrict__ timbres, int n_timbres) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *nt = notes + (size_t)b * max_notes * 5;
    float acc = 0.f;
    if (i < L) {
        const double t = (double)i / (double)sr;
        for (int n = 0; n < max_notes; ++n) {
            const float pitch = nt[n * 5 + 1];
            if (pitch < 0.f) continue;
            const int g = min(max((int)nt[n * 5 + 0], 0), timbres ? n_timbres - 1 : 2);
            const float vel = nt[n * 5 + 2], onset = nt[n * 5 + 3], dur = nt[n * 5 + 4];
            const double tt = t - (double)onset;
            if (tt < 0.0 || tt >= (double)dur + (double)SYN_TAIL) continue;
            Preset pr;
            if (timbres) {
                const float *tb = timbres + (size_t)g * 5;
                pr.H = (int)tb[0]; pr.slope = tb[1]; pr.tau = tb[2]; pr.attack = tb[3]; pr.even = tb[4];
            } else {
                pr = c_presets[g];
            }
            const float ttf = (float)tt;
            float env = fminf(ttf / pr.attack, 1.0f);
            if (pr.tau > 0.f) env *= expf(-ttf / pr.tau);
            env *= expf(-fmaxf(ttf - dur, 0.f) / SYN_RELEASE_TAU);
            const double f0 = 440.0 * exp2(((double)pitch - 69.0) / 12.0);
            float y = 0.f;
            for (int h = 1; h <= pr.H; ++h) {
                if ((double)h * f0 >= 0.5 * (double)sr) break;
                double ph = (double)h * f0 * tt;
                ph -= floor(ph);
                const float wgt = powf((float)h, -pr.slope) * ((h & 1) ? 1.0f : pr.even);
                y += wgt * __builtin_amdgcn_sinf((float)ph);
            }
            const float a = vel * (1.0f / 128.0f);
            acc += (a * a) * (a * a) * y * env;
        }
        wave[(size_t)b * wave_stride + i] = acc;
    }
    float m = block_max(fabsf(acc), red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int *>(peak) + b, __float_as_int(m));
}


// ---- SoundFont 2 sample playback (amt_saga/sf2.py states the definition; oracle/sf2.py is its float64 checker) ----------
// zones [nz][20] f32: key lo, hi, velocity lo, hi, start, end, loop start, loop end (frames in the pool), looped,
// sample rate, root key, scaleTuning (cents / key), tuning (cents), gain, delay, attack, hold, decay (s), sustain (dB),
// release (s).  first [n_prog + 1]: program p plays zones first[p] .. first[p + 1] - 1.  One thread per output sample.
#define SF2_ZF 20
__device__ __forceinline__ float sf2_held_amp(float t, const float *z) {       // envelope while the key is down
    const float ta = t - z[14];
    if (ta < 0.f) return 0.f;
    if (ta < z[15]) return ta / z[15];
    const float td = ta - z[15] - z[16];
    if (td <= 0.f) return 1.0f;
    const float db = fminf(100.0f * td / z[17], z[18]);
    return exp10f(-db * 0.05f);
}
__global__ __launch_bounds__(256) void sf2_synth_kernel(const float *__restrict__ notes, int max_notes,
                                                         const float *__restrict__ samples,
                                                         const float *__restrict__ zones,
                                                         const int32_t *__restrict__ first, int n_prog,
                                                         float *__restrict__ wave, size_t wave_stride, int L, float sr,
                                                         float *__restrict__ peak) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *nt = notes + (size_t)b * max_notes * 5;
    float acc = 0.f;
    if (i < L) {
        const double t = (double)i / (double)sr;
        for (int n = 0; n < max_notes; ++n) {
            const float pitch = nt[n * 5 + 1];
            if (pitch < 0.f) continue;
            const int prog = min(max((int)nt[n * 5 + 0], 0), n_prog - 1);
            const float vel = nt[n * 5 + 2], onset = nt[n * 5 + 3], dur = nt[n * 5 + 4];
            const double tt = t - (double)onset;
            if (tt < 0.0 || tt >= (double)dur + (double)SYN_TAIL) continue;
            const float ttf = (float)tt;
            const int key = (int)pitch, iv = (int)vel;
            float y = 0.f;
            for (int q = first[prog]; q < first[prog + 1]; ++q) {
                const float *z = zones + (size_t)q * SF2_ZF;
                if (key < (int)z[0] || key > (int)z[1] || iv < (int)z[2] || iv > (int)z[3]) continue;
                // volume envelope
                float amp;
                if (ttf < dur) amp = sf2_held_amp(ttf, z);
                else {
                    const float a0 = sf2_held_amp(dur, z);
                    if (a0 <= 1e-5f) continue;
                    const float db = -20.0f * log10f(a0) + 100.0f * (ttf - dur) / z[19];
                    if (db >= 100.0f) continue;
                    amp = exp10f(-db * 0.05f);
                }
                if (amp <= 0.f) continue;
                // playback position
                const double cents = ((double)pitch - (double)z[10]) * (double)z[11] + (double)z[12];
                const double ratio = exp2(cents / 1200.0) * (double)z[9] / (double)sr;
                double pos = (double)z[4] + tt * (double)sr * ratio;
                const double le = (double)z[7], ls = (double)z[6], en = (double)z[5];
                const bool loop = z[8] != 0.f;
                if (loop) { if (pos >= le) pos = ls + fmod(pos - ls, le - ls); }
                else if (pos >= en) continue;
                const double fl = floor(pos);
                const long long i0 = (long long)fl;
                long long i1 = i0 + 1;
                const float fr = (float)(pos - fl);
                const float s0 = samples[i0];
                float s1;
                if (loop) { if (i1 >= (long long)le) i1 = (long long)ls; s1 = samples[i1]; }
                else s1 = i1 < (long long)en ? samples[i1] : 0.f;
                y += (s0 + fr * (s1 - s0)) * (z[13] * amp);
            }
            const float a = vel * (1.0f / 128.0f);
            acc += (a * a) * (a * a) * y;
        }
        wave[(size_t)b * wave_stride + i] = acc;
    }
    float m = block_max(fabsf(acc), red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<int *>(peak) + b, __float_as_int(m));
}

__global__ __launch_bounds__(256) void synth_scale_kernel(const float *__restrict__ notes, int max_notes,
                                                           float *__restrict__ wave, size_t wave_stride, int L,
                                                           const float *__restrict__ peak) {
    const int b = blockIdx.y;
    const float *nt = notes + (size_t)b * max_notes * 5;
    float vmax = 0.f;
    int cnt = 0;
    for (int n = 0; n < max_notes; ++n)
        if (nt[n * 5 + 1] >= 0.f) { vmax = fmaxf(vmax, nt[n * 5 + 2]); ++cnt; }
    if (cnt == 1) vmax = fmaxf(1.f, vmax - 12.f);
    const float pk = peak[b];
    const float a = vmax * (1.0f / 128.0f);
    const float sc = pk > 0.f ? (a * a) * (a * a) / pk : 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256)
        wave[(size_t)b * wave_stride + i] *= sc;
}

// notes[b][0] = {group(program), pitch, velocity or default, 0, dur} from the loop's decisions
__global__ void guess_notes_kernel(const int32_t *__restrict__ program, const int32_t *__restrict__ pitch,
                                   const int32_t *__restrict__ velocity, const int32_t *__restrict__ onset,
                                   const int32_t *__restrict__ end, const int32_t *__restrict__ prog_group,
                                   int n_prog, int n, float frame_seconds, float max_dur, float default_vel,
                                   float *__restrict__ notes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int pr = program ? program[i] : 0;
    pr = pr < 0 ? 0 : (pr >= n_prog ? n_prog - 1 : pr);
    float *o = notes + (size_t)i * 5;
    o[0] = (float)(prog_group ? prog_group[pr] : 0);
    o[1] = (float)pitch[i];
    o[2] = velocity ? (float)velocity[i] : default_vel;
    o[3] = 0.f;
    const int d = max(end[i] - onset[i], 0);
    o[4] = fminf((float)d * frame_seconds, max_dur);
}

extern "C" {

int amt_synth_windows_timbres(const float *notes, int max_notes, int B, int L, float sample_rate,
                              const float *timbres, int n_timbres, float *wave, size_t wave_stride,
                              float *peak_scratch, void *stream) {
    if (!notes || !wave || !peak_scratch || B <= 0 || L <= 0 || max_notes <= 0 || sample_rate <= 0)
        return AMT_E_INVALID;
    if (timbres && n_timbres <= 0) return AMT_E_INVALID;
    if (wave_stride < (size_t)L) return AMT_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    AMT_HIP_CHECK(hipMemsetAsync(peak_scratch, 0, sizeof(float) * B, st));
    dim3 grid((L + 255) / 256, B);
    synth_kernel<<<grid, 256, 0, st>>>(notes, max_notes, wave, wave_stride, L, sample_rate, peak_scratch, timbres, n_timbres);
    int gx = (L + 255) / 256;
    if (gx > 64) gx = 64;
    synth_scale_kernel<<<dim3(gx, B), 256, 0, st>>>(notes, max_notes, wave, wave_stride, L, peak_scratch);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_synth_windows(const float *notes, int max_notes, int B, int L, float sample_rate, float *wave,
                      size_t wave_stride, float *peak_scratch, void *stream) {
    return amt_synth_windows_timbres(notes, max_notes, B, L, sample_rate, nullptr, 0, wave, wave_stride, peak_scratch, stream);
}

int amt_sf2_synth_windows(const float *notes, int max_notes, int B, int L, float sample_rate, const float *samples,
                          int n_samples, const float *zones, int n_zones, const int32_t *first, int n_prog, float *wave,
                          size_t wave_stride, float *peak_scratch, void *stream) {
    if (!notes || !wave || !peak_scratch || !samples || !zones || !first) return AMT_E_INVALID;
    if (B <= 0 || L <= 0 || max_notes <= 0 || sample_rate <= 0 || n_samples <= 0 || n_zones <= 0 || n_prog <= 0)
        return AMT_E_INVALID;
    if (wave_stride < (size_t)L) return AMT_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    AMT_HIP_CHECK(hipMemsetAsync(peak_scratch, 0, sizeof(float) * B, st));
    sf2_synth_kernel<<<dim3((L + 255) / 256, B), 256, 0, st>>>(notes, max_notes, samples, zones, first, n_prog, wave,
                                                               wave_stride, L, sample_rate, peak_scratch);
    int gx = (L + 255) / 256;
    if (gx > 64) gx = 64;
    synth_scale_kernel<<<dim3(gx, B), 256, 0, st>>>(notes, max_notes, wave, wave_stride, L, peak_scratch);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

int amt_guess_notes(const int32_t *program, const int32_t *pitch, const int32_t *velocity,
                    const int32_t *onset, const int32_t *end, const int32_t *prog_group, int n_prog,
                    int n, float frame_seconds, float max_dur, float default_velocity, float *notes,
                    void *stream) {
    if (!pitch || !onset || !end || !notes || n <= 0 || n_prog <= 0) return AMT_E_INVALID;
    guess_notes_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        program, pitch, velocity, onset, end, prog_group, n_prog, n, frame_seconds, max_dur,
        default_velocity, notes);
    AMT_LAUNCH_CHECK();
    return AMT_OK;
}

}  // extern "C"
