"""Oracle (test infrastructure): window-by-window comparison of the product's detect -> subtract loop with
LoopOracle, shared by tests/test_gpu_loop.py, tests/test_gpu_synth.py and __graft_entry__.smoke().

Policy.  The rounded decisions (onset, end, pitch, velocity; argmax of the instrument probabilities) are compared
bit for bit, the pre-rounding head outputs (what res_net.predict returns, RDCNN.py:591-597) to a band per head, the
residual magnitude to 1e-4 of its maximum.  A float closer to a rounding boundary than the distance between the
two arithmetics can land on either side, so a band is unavoidable -- but it is derived from the MEASURED distance
between the product's and the oracle's floats (profiles/r03/loop_float_diffs.json, collected by the GPU tests over
every decision of every case; FLOAT_TOL = ~3 x those maxima), and a window with a decision inside the band is NOT
skipped: the oracle adopts the product's integer for that one decision (only the neighbouring integer across that
boundary is accepted, LoopOracle._round) and everything else of the window -- every float of every iteration, every
other decision, the residual -- is compared as for any other window.  (Round 2 skipped every window with a decision
within 0.02 of a tie: 28 % of the windows left the test uncompared.)

Units: output units of the head (frames, semitones, velocity steps, probability)."""
import numpy as np

# Measured on the GPU (profiles/r03/loop_float_diffs.json: |product float - oracle float| over 632 decisions of the
# loop parity cases, split-fp16 convolutions; the distance includes the feature kernels' f32 differences -- STFT,
# iSTFT, CQT -- not just the heads' arithmetic): timing 1.4e-3 frames max (p99 7e-4), pitch 6.3e-4 semitones, velocity
# 7.0e-4 steps, instrument probabilities 1.2e-5.  The bands are ~3x those maxima.
FLOAT_TOL = {'timing_start': 4e-3, 'timing_end': 4e-3, 'pitch': 2e-3, 'velocity': 2e-3, 'instrument': 1e-4}


def bands_for(p, scale=1.0):
    """Tie / float bands (output units of each head) for Hyperparams p."""
    return {k: v * scale for k, v in FLOAT_TOL.items()}


def compare_windows(orc, waves, refs, events, trace, mags, ref_max, bands, window0=0, n_bins=None, diffs=None):
    """orc: LoopOracle.  waves [B, L] numpy; refs: dict name -> [B] numpy; events [iters, B, 7] numpy (product);
    trace: list (per iteration) of dict head -> numpy [B, K] (product's pre-rounding outputs); mags [B, T, ldf]
    numpy (product's residual, frame-major); ref_max [B].  Asserts per window; returns (clean, ties, forced):
    windows without / with a decision inside its band, and decisions handed to the oracle."""
    B = waves.shape[0]
    clean = ties = forced = 0
    for i in range(B):
        r = {k: float(v[i]) for k, v in refs.items()}
        ev_ref, mag_ref = orc.run_window(waves[i], r, window0 + i, force=(events[:, i, :], bands))
        near = False
        for name, it, y, margin, v, was_forced in orc.decisions:
            g = trace[it][name][i]
            d = float(np.abs(np.asarray(g, np.float64).ravel() - np.asarray(y, np.float64).ravel()).max())
            if diffs is not None:
                diffs.setdefault(name, []).append(d)
            assert d <= bands[name], ('float', i, name, it, d, bands[name])
            near |= margin < bands[name]
            forced += was_forced
        ties += near
        clean += not near
        assert np.array_equal(events[:, i, :], ev_ref), ('events', i, events[:, i, :], ev_ref)
        F = mag_ref.shape[0]
        mag = mags[i][:, :F].T
        assert np.abs(mag - mag_ref).max() / max(mag_ref.max(), 1e-30) < 1e-4, ('residual', i)
        assert abs(float(ref_max[i]) - mag_ref.max()) / max(mag_ref.max(), 1e-30) < 1e-4, ('ref_max', i)
    return clean, ties, forced
