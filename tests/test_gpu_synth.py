"""GPU: the HIP note synthesiser (amt_synth_windows) against its float64 CPU
specification (oracle/synth.py:render_window), and the loop's 'render' guess
mode (one synthesised guess per decision, training.py:421-431) against the
oracle loop given the CPU synth as its guess function."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    assert torch.cuda.is_available()
    from amt_saga import synth, loop, hyperparams
    from oracle import loop as oloop
    return dict(torch=torch, synth=synth, loop=loop, hp=hyperparams, oloop=oloop)


from oracle import synth as osynth      # noqa: E402  (the float64 CPU restatement of the synth definition)


def test_synth_kernel_vs_cpu_definition(env):
    synth, torch = env['synth'], env['torch']
    rng = np.random.default_rng(5)
    L, sr = 3 * 44100 + 77, 44100
    notes = [synth.random_notes(rng, n, groups=(0, 1, 2), max_onset=1.5) for n in (1, 1, 2, 3, 5)]
    notes.append([(1, 108, 127, 0.0, 0.5)])             # top pitch: harmonics cut at Nyquist
    notes.append([(0, 21, 5, 2.9, 2.0)])                # quiet, late note running out of the window
    notes.append([(2, 60, 64, 0.25, 0.1), (2, 60, 64, 0.25, 0.1)])   # identical notes
    got = synth.render_windows_device(notes, L, sr).cpu().numpy()
    for i, ns in enumerate(notes):
        want = osynth.render_window(ns, L, sr).numpy()
        peak = np.abs(want).max()
        assert peak > 0
        assert np.abs(got[i] - want).max() / peak < 2e-5, i
        vmax = max(n[2] for n in ns)
        vmax = max(1, vmax - 12) if len(ns) == 1 else vmax
        assert abs(np.abs(got[i]).max() - (vmax / 128.0) ** 4) / (vmax / 128.0) ** 4 < 1e-5
    # unused slots / all-silent window
    nt = synth.notes_tensor([[(0, 60, 100, 4.0, 0.5)], []], max_notes=3)
    w = synth.render_windows_device(torch.from_numpy(nt).cuda(), 4410, sr).cpu().numpy()
    assert np.all(w == 0)
    # deterministic
    a = synth.render_windows_device(notes, L, sr)
    assert torch.equal(a, synth.render_windows_device(notes, L, sr))


def test_synth_gm_program_timbres_vs_cpu_definition(env):
    """Per-program timbres: every General MIDI program 0..127 through amt_synth_windows_timbres with the
    build's table against the float64 restatement; mixtures of programs in one window; programs 0 / 24 / 40 equal the
    three built-in groups bit for bit; a caller's own table."""
    synth, torch = env['synth'], env['torch']
    sr, L = 44100, 44100 + 300
    singles = [[(pr, 45 + (pr * 7) % 40, 40 + (pr * 5) % 80, 0.05, 0.4)] for pr in range(128)]
    rng = np.random.default_rng(3)
    mixes = [[(int(rng.integers(0, 128)), int(rng.integers(30, 100)), int(rng.integers(20, 127)),
               float(rng.uniform(0, 0.5)), float(rng.uniform(0.1, 0.6))) for _ in range(n)] for n in (2, 3, 6)]
    notes = singles + mixes
    got = synth.render_windows_device(notes, L, sr, timbres='gm').cpu().numpy()
    distinct = set()
    for i, ns in enumerate(notes):
        want = osynth.render_window(ns, L, sr, timbres='gm').numpy()
        peak = np.abs(want).max()
        assert peak > 0 and np.abs(got[i] - want).max() / peak < 2e-5, i
        if i < 128:
            distinct.add(tuple(np.round(synth.gm_timbre_table()[i], 6)))
    assert len(distinct) == 128                                   # no two programs share a timbre
    for prog, grp in ((0, 0), (24, 2), (40, 1)):
        a = synth.render_windows_device([[(prog, 60, 90, 0.1, 0.5)]], L, sr, timbres='gm')
        b = synth.render_windows_device([[(grp, 60, 90, 0.1, 0.5)]], L, sr)
        assert torch.equal(a, b), prog
    own = np.array([[3, 2.0, 0.0, 0.01, 0.0]], np.float32)         # three harmonics, the even one silent
    w = synth.render_windows_device([[(0, 69, 100, 0.0, 0.5)]], L, sr, timbres=own).cpu().numpy()[0]
    spec = np.abs(np.fft.rfft(w[:22050] * np.hanning(22050)))
    f = np.fft.rfftfreq(22050, 1 / sr)
    pk = lambda hz: spec[np.abs(f - hz) < 6].max()
    assert pk(880.0) < 1e-3 * pk(440.0) and pk(1320.0) > 0.05 * pk(440.0)
    with pytest.raises(ValueError):
        synth.render_windows_device(singles[:1], L, sr, timbres=np.zeros((2, 4), np.float32))


def test_sf2_playback_kernel_vs_cpu_definition(env):
    """SoundFont 2 sample playback (optional soundfont path): amt_sf2_synth_windows on the font written by
    tests/sf2_fixture.py against oracle/sf2.py -- looped and one-shot samples, key and velocity splits, layered zones
    with a preset-level transposition, envelopes, notes outside every zone, mixtures; and the loop's render mode
    playing its guesses from the font."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sf2_fixture
    from amt_saga import sf2
    from oracle import sf2 as osf2
    torch = env['torch']
    font = sf2.SoundFont(sf2_fixture.build())
    sr, L = 44100, 2 * 44100 + 123
    notes = [[(0, 81, 100, 0.0, 1.5)], [(0, 60, 64, 0.25, 0.3)], [(24, 57, 90, 0.0, 1.2)], [(24, 45, 120, 0.1, 0.5)],
             [(40, 64, 80, 0.05, 0.6)], [(40, 90, 127, 0.0, 0.4)], [(100, 20, 90, 0.0, 0.2), (7, 60, 90, 0.0, 0.2)],
             [(0, 72, 50, 0.3, 0.2), (24, 69, 101, 0.0, 0.7), (40, 73, 110, 0.5, 1.0)]]
    got = sf2.render_windows_device(notes, L, font, sr).cpu().numpy()
    for i, ns in enumerate(notes):
        want = osf2.render_window(ns, L, font.samples, font.programs, sr)
        peak = np.abs(want).max()
        if i == 6:
            assert peak == 0 and np.all(got[i] == 0)               # programs the font has no preset for
            continue
        assert peak > 0 and np.abs(got[i] - want).max() / peak < 5e-5, (i, np.abs(got[i] - want).max() / peak)
    a = sf2.render_windows_device(notes, L, font, sr)
    assert torch.equal(a, sf2.render_windows_device(notes, L, font, sr))
    # the loop plays its guesses from the font (programs without a preset are silent guesses: nothing is subtracted)
    p = env['hp'].Hyperparams(N=2048, window_size_note_time=1)
    lp = env['loop'].TranscriptionLoop(p, heads=('timing', 'pitch', 'velocity'), iters=1, guess='render',
                                       soundfont=font).setup_device()
    wave, _ = env['synth'].make_windows(4, p.H * (p.timing_frames - 1), seed=3, notes_per_window=(1, 2), max_onset=0.4,
                                        device='cuda')
    events, b = lp.run(wave)
    assert events.shape[1] == 4 and torch.isfinite(b.mag).all()
    with pytest.raises(ValueError):
        env['loop'].TranscriptionLoop(p, guess='bank', soundfont=font)


def test_synth_argument_checks(env):
    from amt_saga import _lib
    lib = _lib.load()
    assert lib.amt_synth_windows(None, 1, 1, 10, 44100.0, None, 10, None, None) == _lib.AMT_E_INVALID
    torch = env['torch']
    nt = torch.zeros(1, 1, 5, device='cuda')
    w = torch.zeros(1, 8, device='cuda')
    pk = torch.zeros(1, device='cuda')
    assert lib.amt_synth_windows(nt.data_ptr(), 1, 1, 16, 44100.0, w.data_ptr(), 8, pk.data_ptr(),
                                 None) == _lib.AMT_E_SHAPE


@pytest.mark.parametrize('heads,iters,seeds,timbres', [
    (('timing', 'pitch', 'velocity'), 2, None, None),
    (('timing', 'pitch', 'instrument', 'velocity'), 2, None, None),
    (('timing', 'pitch', 'instrument', 'velocity'), 2, None, 'gm'),     # every decided program with its own timbre
])
def test_loop_render_guess_vs_oracle(env, heads, iters, seeds, timbres):
    torch, synth = env['torch'], env['synth']
    p = env['hp'].Hyperparams(N=2048, window_size_note_time=1)
    groups = (0, 1, 2) if 'instrument' in heads else (0,)
    lp = env['loop'].TranscriptionLoop(p, heads=heads, iters=iters, groups=groups, seeds=seeds,
                                       guess='render', timbres=timbres).setup_device()
    L = p.H * (p.timing_frames - 1)
    B = 6
    wave, _ = synth.make_windows(B, L, seed=22, notes_per_window=(1, 3), groups=groups,
                                 max_onset=0.4, device='cuda')
    lp.trace = []
    events, b = lp.run(wave, window0=7)
    trace, lp.trace = [{k: v.cpu().numpy() for k, v in t.items()} for t in lp.trace], None
    ev = events.cpu().numpy()
    table = synth.prog_group_table(p.instrument_classes)
    Lg = lp.bank_len

    def guess_fn(program, pitch, velocity, frames):
        dur = min(float(np.float32(frames) * np.float32(p.H / p.sr)), 1.0)
        vel = velocity if velocity > 0 else 100
        g = int(program) if timbres == 'gm' else int(table[program])
        return osynth.render_window([(g, pitch, vel, 0.0, dur)], Lg, p.sr, timbres=timbres).numpy()

    orc = env['oloop'].LoopOracle(p, heads, {k: n.weights for k, n in lp.nets.items()}, iters=iters,
                                  guess_fn=guess_fn)
    from oracle.compare import bands_for, compare_windows
    clean, ties, forced = compare_windows(orc, wave.cpu().numpy(), {k: v.cpu().numpy() for k, v in lp.refs.items()},
                                          ev, trace, b.mag.cpu().numpy(), b.ref_max.cpu().numpy(), bands_for(p),
                                          window0=7)
    assert clean + ties == B and clean >= B - 1
