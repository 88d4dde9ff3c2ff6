"""Oracle (test infrastructure): float64 CPU restatement of the build's additive synthesiser -- the
stand-in for note_sequence.render() (fluidsynth + GM soundfont, /root/reference/util_audio.py:758-786,
neither of which exists here; SURVEY 7 hard part 3, 8d).  The definition (presets, envelopes, the
amplitude law of render(), util_audio.py:778-781) is stated in amt-saga_amd/amt_saga/synth.py; this file
evaluates it in float64 and is what tests/ compare the HIP kernel amt_synth_windows with and what the CPU
baseline of bench.py renders its guess bank with.  Only the amplitude law is pinned by the reference; the
timbres are the build's (parity unpinned vs fluidsynth)."""
import numpy as np
import torch

PRESETS = {
    'piano': dict(H=12, slope=1.5, tau=0.6, attack=0.002, sustain=False),
    'strings': dict(H=16, slope=1.0, tau=None, attack=0.08, sustain=True),
    'guitar': dict(H=10, slope=1.2, tau=0.35, attack=0.002, sustain=False),
}
PROGRAM_GROUPS = ['piano', 'strings', 'guitar']
RELEASE_TAU = 0.06
TAIL_SECONDS = 1.0


# per-program timbres (amt-saga_amd/amt_saga/synth.py: GM_FAMILIES / gm_timbre): sixteen General MIDI families, the
# eight programs of a family step slope (+0.04) and decay (-5 %); restated here, not imported
_GM = [(12, 1.50, 0.60, 0.002, 1.0), (6, 1.00, 0.25, 0.001, 1.0), (9, 0.80, None, 0.010, 1.0), (10, 1.20, 0.35, 0.002, 1.0),
       (8, 1.30, 0.50, 0.004, 1.0), (16, 1.00, None, 0.080, 1.0), (14, 1.10, None, 0.120, 1.0), (14, 0.90, None, 0.030, 1.0),
       (12, 1.00, None, 0.020, 0.3), (5, 1.80, None, 0.040, 0.5), (20, 1.00, None, 0.005, 1.0), (10, 1.40, None, 0.250, 1.0),
       (12, 1.20, 1.20, 0.100, 1.0), (10, 1.10, 0.30, 0.003, 1.0), (6, 0.90, 0.15, 0.001, 1.0), (8, 1.00, 0.40, 0.010, 1.0)]


def gm_preset(program):
    H, slope, tau, attack, even = _GM[int(program) // 8]
    i = int(program) % 8
    f32 = lambda v: float(np.float32(v))              # the kernel reads the table as float32
    return dict(H=H, slope=f32(slope + 0.04 * i), tau=None if tau is None else f32(tau * (1.0 - 0.05 * i)),
                attack=f32(attack), even=f32(even))


def _note(t, group, pitch, onset, dur, sr, timbres=None):
    """t: [L] float64 tensor of seconds; returns float64 [L].  timbres = 'gm': `group` is a General MIDI program."""
    pr = gm_preset(group) if timbres == 'gm' else dict(PRESETS[PROGRAM_GROUPS[group]], even=1.0)
    f0 = 440.0 * 2.0 ** ((pitch - 69) / 12.0)
    tt = t - onset
    on = (tt >= 0).to(t.dtype)
    ttc = torch.clamp(tt, min=0.0)
    env = torch.clamp(ttc / pr['attack'], max=1.0)
    if pr['tau'] is not None:
        env = env * torch.exp(-ttc / pr['tau'])
    rel = torch.clamp(tt - dur, min=0.0)
    env = env * torch.exp(-rel / RELEASE_TAU) * on
    env = env * (tt < dur + TAIL_SECONDS).to(t.dtype)
    y = torch.zeros_like(t)
    for h in range(1, pr['H'] + 1):
        if h * f0 >= sr / 2:
            break
        y = y + (h ** -pr['slope']) * (1.0 if h % 2 else pr['even']) * torch.sin(2.0 * np.pi * h * f0 * ttc)
    return y * env


def render_window(notes, L, sr=44100, device='cpu', timbres=None):
    """notes: list of (group, pitch, velocity, onset_s, dur_s).  float32 [L]."""
    t = torch.arange(L, dtype=torch.float64, device=device) / sr
    wf = torch.zeros(L, dtype=torch.float64, device=device)
    for (g, p, v, t0, d) in notes:
        # note records are float32 (the [B, M, 5] tensor the kernel reads): onset and duration
        # take their float32 values here too, so both implementations see the same note
        t0, d = float(np.float32(t0)), float(np.float32(d))
        amp = (v / 128.0) ** 4          # fluidsynth-like loudness spread between notes
        wf = wf + amp * _note(t, g, p, t0, d, sr, timbres)
    vel_max = max(n[2] for n in notes)
    if len(notes) == 1:
        vel_max = max(1, vel_max - 12)
    peak = wf.abs().max()
    if float(peak) > 0:
        wf = wf * ((vel_max / 128.0) ** 4 / peak)
    return wf.to(torch.float32)




def render_notes(note_lists, L, sr=44100, timbres=None):
    """[B, L] float32 numpy array for a list of note lists."""
    return torch.stack([render_window(ns, L, sr, timbres=timbres) for ns in note_lists]).numpy()


def guess_bank_waves(groups=(0,), pitch_lo=21, pitch_hi=108, dur=1.0, velocity=100, sr=44100):
    """One rendered single-note guess per (group, pitch): [G*n_pitch, L_g] float32,
    L_g = (dur + 1 s tail) * sr samples (the reference's guess = note + 1 s, util_audio.py:876)."""
    Lg = int(round((dur + TAIL_SECONDS) * sr))
    notes = [[(g, p, velocity, 0.0, dur)] for g in groups for p in range(pitch_lo, pitch_hi + 1)]
    return render_notes(notes, Lg, sr)
