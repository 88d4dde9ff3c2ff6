"""Deterministic additive synthesiser: the stand-in for note_sequence.render()
(fluidsynth + GM soundfont, /root/reference/util_audio.py:758-786), neither of
which exists here (SURVEY 7 hard part 3, 8d).  It produces (a) the synthetic
benchmark windows and (b) the guess-template bank the subtraction step reads.

    note(g, p, v, t0, d) = env_g(t - t0; d) * sum_{h=1..H_g} a_{g,h} sin(2 pi h f_p (t - t0))
    f_p = 440 * 2**((p-69)/12); harmonics above Nyquist are dropped
    presets: piano   H=12, a_h = h**-1.5, exp decay tau = 0.6 s
             strings H=16, a_h = h**-1.0, 80 ms linear attack, sustain
             guitar  H=10, a_h = h**-1.2, exp decay tau = 0.35 s
    every note is followed by a release (exp, 60 ms) inside a 1 s tail (:876)
    window scaling follows render() (:778-781):
        wf * (vel_max/128)**4 / max|wf|,  vel_max - 12 for a single note

This module holds the definition's constants, the note-list generation (host integers) and the
device renderer render_windows_device() = the HIP kernel amt_synth_windows (csrc/amt_synth.hip): what
the benchmark, the guess bank and the loop's "render" guess mode run.  It raises without a GPU like
every other product path.  The float64 CPU restatement of the same definition, used as the kernel's
checker and by the CPU baseline, lives in oracle/synth.py.
"""
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


def program_to_group(program):
    """GM program number -> preset index: 24-31 guitars, 40-55 strings/ensemble,
    everything else piano."""
    if 24 <= program <= 31:
        return 2
    if 40 <= program <= 55:
        return 1
    return 0


def prog_group_table(n_prog=112):
    return np.array([program_to_group(p) for p in range(n_prog)], dtype=np.int32)


# ---- per-program timbres ------------------------------------------------------------------------------
# The reference renders every note through the soundfont preset of its MIDI program (util_audio.py:758-786); without a
# soundfont the build gives each of the 128 General MIDI programs its own additive timbre: the sixteen GM families set
# (harmonics, spectral slope, decay, attack, even-harmonic weight), the eight programs of a family move slope and decay
# in steps.  Programs 0, 24 and 40 are exactly the three groups above.  Build-defined, like those.
GM_FAMILIES = [
    # H, slope, tau (None = sustained), attack, even      GM programs
    (12, 1.50, 0.60, 0.002, 1.0),      # 0-7     piano
    (6, 1.00, 0.25, 0.001, 1.0),       # 8-15    chromatic percussion
    (9, 0.80, None, 0.010, 1.0),       # 16-23   organ
    (10, 1.20, 0.35, 0.002, 1.0),      # 24-31   guitar
    (8, 1.30, 0.50, 0.004, 1.0),       # 32-39   bass
    (16, 1.00, None, 0.080, 1.0),      # 40-47   strings
    (14, 1.10, None, 0.120, 1.0),      # 48-55   ensemble
    (14, 0.90, None, 0.030, 1.0),      # 56-63   brass
    (12, 1.00, None, 0.020, 0.3),      # 64-71   reed
    (5, 1.80, None, 0.040, 0.5),       # 72-79   pipe
    (20, 1.00, None, 0.005, 1.0),      # 80-87   synth lead
    (10, 1.40, None, 0.250, 1.0),      # 88-95   synth pad
    (12, 1.20, 1.20, 0.100, 1.0),      # 96-103  synth effects
    (10, 1.10, 0.30, 0.003, 1.0),      # 104-111 ethnic
    (6, 0.90, 0.15, 0.001, 1.0),       # 112-119 percussive
    (8, 1.00, 0.40, 0.010, 1.0),       # 120-127 sound effects
]


def gm_timbre(program):
    """(H, slope, tau or 0.0 for sustained, attack, even) of General MIDI program 0..127."""
    H, slope, tau, attack, even = GM_FAMILIES[int(program) // 8]
    i = int(program) % 8
    return (H, slope + 0.04 * i, 0.0 if tau is None else tau * (1.0 - 0.05 * i), attack, even)


def gm_timbre_table(n_prog=128):
    """float32 [n_prog, 5]: the table amt_synth_windows_timbres reads (note field 0 = MIDI program)."""
    return np.array([gm_timbre(p) for p in range(n_prog)], dtype=np.float32)


def notes_tensor(notes, max_notes=None):
    """list (per window) of note lists -> float32 [B, max_notes, 5]; unused slots pitch -1."""
    mn = max_notes or max(len(n) for n in notes)
    out = np.zeros((len(notes), mn, 5), dtype=np.float32)
    out[:, :, 1] = -1.0
    for i, ns in enumerate(notes):
        for j, n in enumerate(ns):
            out[i, j] = n
    return out


_TIMBRE_DEV = {}


def render_windows_device(notes, L, sr=44100, out=None, timbres=None):
    """HIP synthesiser.  notes: list of note lists, or a device float32 [B, M, 5] tensor
    (rows {group, pitch, velocity, onset_s, dur_s}; pitch < 0 = unused).  Returns a device
    float32 [B, L] tensor.  timbres: None = field 0 is one of the three groups; 'gm' = field 0 is a General MIDI
    program rendered with gm_timbre_table(); or a float32 [n, 5] table of the caller's."""
    from . import _lib
    from .device import empty, ptr, stream_ptr, to_dev
    lib = _lib.load()
    nt = notes if isinstance(notes, torch.Tensor) else to_dev(notes_tensor(notes))
    if nt.dim() == 2:
        nt = nt[:, None, :]
    assert nt.dtype == torch.float32 and nt.shape[-1] == 5 and nt.is_contiguous()
    B, M = nt.shape[0], nt.shape[1]
    wave = out if out is not None else empty((B, int(L)))
    peak = empty((B,))
    if timbres is None:
        _lib.check(lib.amt_synth_windows(ptr(nt), M, B, int(L), float(sr), ptr(wave), wave.stride(0),
                                         ptr(peak), stream_ptr()))
        return wave
    if isinstance(timbres, str):
        if timbres != 'gm':
            raise ValueError("timbres: None, 'gm' or a [n, 5] table")
        if 'gm' not in _TIMBRE_DEV:
            _TIMBRE_DEV['gm'] = to_dev(gm_timbre_table())
        tb = _TIMBRE_DEV['gm']
    else:
        tb = timbres if isinstance(timbres, torch.Tensor) else to_dev(np.ascontiguousarray(timbres, dtype=np.float32))
    if tb.dim() != 2 or tb.shape[1] != 5 or tb.dtype != torch.float32 or not tb.is_contiguous():
        raise ValueError('timbre table must be float32 [n, 5]')
    _lib.check(lib.amt_synth_windows_timbres(ptr(nt), M, B, int(L), float(sr), ptr(tb), int(tb.shape[0]), ptr(wave),
                                             wave.stride(0), ptr(peak), stream_ptr()))
    return wave


def _on_gpu(device):
    return torch.device(device).type == 'cuda'


def random_notes(rng, n_notes, groups=(0,), max_onset=3.0):
    """p ~ U{21..108}, v ~ U{5..125}, t0 ~ U[0,3), d ~ U[0.1,2]."""
    out = []
    for _ in range(n_notes):
        out.append((int(rng.choice(groups)), int(rng.integers(21, 109)), int(rng.integers(5, 126)),
                    float(rng.uniform(0, max_onset)), float(rng.uniform(0.1, 2.0))))
    return out


def window_notes(B, seed, notes_per_window=(3, 3), groups=(0,), max_onset=3.0):
    """The seeded note lists of B synthetic windows (host integers / floats only)."""
    rng = np.random.default_rng(seed)
    notes = []
    for _ in range(B):
        n = int(rng.integers(notes_per_window[0], notes_per_window[1] + 1))
        notes.append(random_notes(rng, n, groups, max_onset))
    return notes


def bank_notes(groups=(0,), pitch_lo=21, pitch_hi=108, dur=1.0, velocity=100):
    """One single-note guess per (group, pitch) -- the guess-template bank."""
    return [[(g, p, velocity, 0.0, dur)] for g in groups for p in range(pitch_lo, pitch_hi + 1)]


def make_windows(B, L, seed, notes_per_window=(3, 3), groups=(0,), sr=44100, device='cuda',
                 max_onset=3.0):
    """[B, L] float32 device tensor of synthetic windows + the note lists."""
    notes = window_notes(B, seed, notes_per_window, groups, max_onset)
    if not _on_gpu(device):
        raise RuntimeError('amt_saga.synth renders on the GPU only (the CPU restatement is oracle/synth.py)')
    return render_windows_device(notes, L, sr), notes


def guess_bank_waves(groups=(0,), pitch_lo=21, pitch_hi=108, dur=1.0, velocity=100, sr=44100,
                     device='cuda'):
    """One rendered single-note guess per (group, pitch): [G*n_pitch, L_g] float32,
    L_g = (dur + 1 s tail) * sr samples (the reference's guess = note + 1 s, :876)."""
    Lg = int(round((dur + TAIL_SECONDS) * sr))
    notes = bank_notes(groups, pitch_lo, pitch_hi, dur, velocity)
    if not _on_gpu(device):
        raise RuntimeError('amt_saga.synth renders on the GPU only (the CPU restatement is oracle/synth.py)')
    return render_windows_device(notes, Lg, sr)
