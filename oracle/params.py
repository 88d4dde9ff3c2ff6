"""Oracle (test infrastructure): restatement of util_train_test.py:15-146
(Hyperparams, check_shape, list_to_nd_array).  Pinned by
tests/golden/reference_vectors.npz (emitted by the reference's own module)."""
import numpy as np


class HyperparamsOracle:
    """util_train_test.py:15-79 (only the fields the hot path reads)."""

    def __init__(self, N=4096, sr=44100, H=None, window_size_note_time=None,
                 bins_per_tone=4, batch_size=8):
        self.N = N
        self.sr = sr
        self.H = int(N / 4) if H is None else H
        self.window_size_note_time = 6 if window_size_note_time is None else window_size_note_time
        self.convolutional_layer_count = 33
        self.pool_layer_frequency = 12
        self.feature_expand_frequency = 12
        self.residual_layer_frequencies = [2]
        self.timing_frames = int(self.window_size_note_time * self.sr / self.H)
        self.timing_bands = max(20, 20 * bins_per_tone // 6)
        self.kernel_size_timing = [(4, 16)]
        self.pool_size_timing = [(int(2 * max(1, np.log2(bins_per_tone // 2))), 8)]
        self.pitch_frames = 8
        self.pitch_low = 21
        self.pitch_high = 108
        self.pitch_bins_per_tone = max(1, bins_per_tone // 2)
        self.pitch_bands = (self.pitch_high - self.pitch_low) * self.pitch_bins_per_tone
        self.kernel_size_pitch = [(4, 2)]
        self.pool_size_pitch = [(4, 2)]
        self.instrument_frames = self.pitch_frames
        self.instrument_bins_per_tone = bins_per_tone
        self.instrument_bands = self.instrument_bins_per_tone * (self.pitch_high - self.pitch_low)
        self.instrument_classes = 112
        self.kernel_size_instrument = [(4, 2)]
        self.pool_size_instrument = [(int(4 * max(1, np.log2(bins_per_tone))), 2)]
        self.bins_velocity = 36
        self.velocity_min = 5
        self.velocity_max = 125
        self.kernel_size_velocity = [(2, 2)]
        self.pool_size_velocity = [(2, 2)]
        self.batch_size = batch_size


def check_shape(spec, bands, frames):                      # util_train_test.py:93-112
    if isinstance(spec, (list, tuple)):
        if isinstance(spec[0], (list, tuple)):
            spec_shape = spec[0][0].shape
        else:
            spec_shape = spec[0].shape
    else:
        spec_shape = spec.shape
    if spec_shape != (bands, frames):
        raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format(
            (bands, frames), spec_shape))


def list_to_nd_array(spec, label):                         # util_train_test.py:114-146
    if isinstance(spec, (list, tuple)):
        if isinstance(spec[0], (list, tuple)):
            n_tow = len(spec[0])
            xs = [np.stack([np.asarray(sp[i], dtype=np.float64) for sp in spec])[..., None]
                  for i in range(n_tow)]
            ys = None if label is None else np.asarray(
                [[l] for l in label], dtype=np.float64).reshape(-1, 1)
            return xs, ys
        x = np.stack([np.asarray(s, dtype=np.float64) for s in spec])[..., None]
        ys = None if label is None else np.asarray(
            [[l] for l in label], dtype=np.float64).reshape(-1, 1)
        return x, ys
    x = np.asarray(spec)[np.newaxis, :, :, np.newaxis]
    ys = None if label is None else np.expand_dims(label, axis=0)
    return x, ys
