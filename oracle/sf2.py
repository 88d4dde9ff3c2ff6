"""Oracle (test infrastructure): float64 restatement of the build's SoundFont 2 sample playback -- the definition in
amt-saga_amd/amt_saga/sf2.py (the optional soundfont path of SURVEY 8f-1; the reference plays notes through fluidsynth,
/root/reference/util_audio.py:758-786, :819-936, which does not exist here: parity with fluidsynth is unpinned).  Takes
plain data (the sample pool and flattened zone dicts); tests/ compare the HIP kernel amt_sf2_synth_windows with it."""
import numpy as np

TAIL_SECONDS = 1.0


def _held_amp(t, z):
    ta = t - z['delay']
    if ta < 0:
        return 0.0
    if ta < z['attack']:
        return ta / z['attack']
    td = ta - z['attack'] - z['hold']
    if td <= 0:
        return 1.0
    return 10.0 ** (-min(100.0 * td / z['decay'], z['sustain_db']) / 20.0)


def _zone_wave(samples, z, pitch, tt, dur, sr):
    """tt: float64 [n] seconds since note on (>= 0, < dur + tail).  Returns float64 [n]."""
    f32 = lambda v: float(np.float32(v))                    # the kernel reads the zone table as float32
    zz = {k: (f32(v) if isinstance(v, float) else v) for k, v in z.items()}
    out = np.zeros(len(tt))
    a0 = _held_amp(f32(dur), zz)
    cents = (pitch - zz['root']) * zz['scale'] + zz['tune']
    ratio = 2.0 ** (cents / 1200.0) * zz['rate'] / sr
    pos = zz['start'] + tt * sr * ratio
    ls, le, en = zz['loop_start'], zz['loop_end'], zz['end']
    for j in range(len(tt)):
        t = float(tt[j])
        if t < dur:
            amp = _held_amp(t, zz)
        else:
            if a0 <= 1e-5:
                continue
            db = -20.0 * np.log10(a0) + 100.0 * (t - dur) / zz['release']
            if db >= 100.0:
                continue
            amp = 10.0 ** (-db / 20.0)
        if amp <= 0:
            continue
        p = pos[j]
        if zz['loop']:
            if p >= le:
                p = ls + np.fmod(p - ls, le - ls)
        elif p >= en:
            continue
        i0 = int(np.floor(p))
        fr = p - i0
        i1 = i0 + 1
        if zz['loop']:
            s1 = samples[ls if i1 >= le else i1]
        else:
            s1 = samples[i1] if i1 < en else 0.0
        out[j] = (samples[i0] + fr * (s1 - samples[i0])) * zz['gain'] * amp
    return out


def render_window(notes, L, samples, programs, sr=44100):
    """notes: list of (program, pitch, velocity, onset_s, dur_s); programs: {program: [zone dicts]}.  float32 [L]."""
    samples = np.asarray(samples, dtype=np.float64)
    t = np.arange(L, dtype=np.float64) / sr
    wf = np.zeros(L)
    for (prog, pitch, vel, t0, d) in notes:
        t0, d = float(np.float32(t0)), float(np.float32(d))
        tt = t - t0
        sel = np.nonzero((tt >= 0) & (tt < d + TAIL_SECONDS))[0]
        y = np.zeros(len(sel))
        for z in programs.get(int(prog), []):
            if z['key'][0] <= pitch <= z['key'][1] and z['vel'][0] <= vel <= z['vel'][1]:
                y += _zone_wave(samples, z, pitch, tt[sel], d, sr)
        wf[sel] += (vel / 128.0) ** 4 * y
    vel_max = max(n[2] for n in notes)
    if len(notes) == 1:
        vel_max = max(1, vel_max - 12)
    peak = np.abs(wf).max()
    if peak > 0:
        wf = wf * ((vel_max / 128.0) ** 4 / peak)
    return wf.astype(np.float32)
