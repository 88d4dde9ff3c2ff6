"""Configuration bag and input-contract helpers of the classifier heads.

Mirrors /root/reference/util_train_test.py: ``Hyperparams`` (:15-79, same field
names and derivations), ``check_shape`` (:93-112, same ValueError text) and
``list_to_nd_array`` (:114-146, same output shapes/dtypes).  Differences, both
deliberate:
  * works on numpy >= 1.24 (the reference uses the removed ``np.int``);
  * ``list_to_nd_array`` stacks once (the reference re-concatenates per
    sample, O(B^2)) and accepts ``label=None`` for lists, which the reference's
    zip() cannot (its predict path was never exercised).
"""
import math

import numpy as np

PATH_MODEL_META = 'model_meta'
PATH_NOTES = 'notes'
PATH_CHECKPOINTS = 'checkpoints'


def _log2(v):
    # np.log2(0) is -inf in the reference (util_train_test.py:42 with bins_per_tone < 2)
    return math.log2(v) if v > 0 else -math.inf


class Hyperparams:
    def __init__(self, path_data=None, sf_path=None, path_output='.',
                 N=4096, sr=44100, H=None, window_size_note_time=None,
                 models_to_train=(0, 1, 2, 3, 4, 5, 6),
                 bins_per_tone=4,
                 batch_size=8, synth_worker_count=1,
                 parallel_train=False,
                 checkpoint_dir='./data/checkpoints',
                 checkpoint_frequency=200,
                 note_save_freq=0,
                 autoload=False,
                 use_precise_note_count=True):
        self.N = N
        self.sr = sr
        self.H = int(N / 4) if H is None else H
        self.window_size_note_time = 6 if window_size_note_time is None else window_size_note_time
        self.models_to_train = list(models_to_train)

        self.convolutional_layer_count = 33
        self.pool_layer_frequency = 12
        self.feature_expand_frequency = 12
        self.residual_layer_frequencies = [2]

        self.timing_frames = int(self.window_size_note_time * self.sr / self.H)
        self.timing_bands = max(20, 20 * bins_per_tone // 6)
        self.kernel_size_timing = [(4, 16)]
        self.pool_size_timing = [(int(2 * max(1, _log2(bins_per_tone // 2))), 8)]

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
        self.pool_size_instrument = [(int(4 * max(1, _log2(bins_per_tone))), 2)]

        self.bins_velocity = 36
        self.velocity_min = 5
        self.velocity_max = 125
        self.kernel_size_velocity = [(2, 2)]
        self.pool_size_velocity = [(2, 2)]

        self.batch_size = batch_size
        self.checkpoint_dir = checkpoint_dir
        self.checkpoint_frequency = checkpoint_frequency
        self.parallel_train = parallel_train
        self.synth_worker_count = synth_worker_count
        self.path_data = path_data
        self.sf_path = sf_path
        self.path_output = path_output
        self.note_save_freq = note_save_freq
        self.autoload = autoload
        self.use_precise_note_count = use_precise_note_count


def _shape_of(spec):
    if isinstance(spec, (list, tuple)):
        first = spec[0]
        if isinstance(first, (list, tuple)):
            return tuple(first[0].shape)
        return tuple(first.shape)
    return tuple(spec.shape)


def check_shape(spec, bands, frames):
    """Raises ValueError('Invalid Input shape. Expected: (b, f) . Got: (..)')."""
    spec_shape = _shape_of(spec)
    if spec_shape != (bands, frames):
        raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format(
            (bands, frames), spec_shape))


def _labels(label, n):
    if label is None:
        return None
    return np.asarray(list(label), dtype=np.float64).reshape(n, 1)


def list_to_nd_array(spec, label):
    """list of [bands, frames] -> float64 (B, bands, frames, 1) (a list of such
    for multi-tower inputs) and labels (B, 1)."""
    if isinstance(spec, (list, tuple)):
        if isinstance(spec[0], (list, tuple)):
            n_tow = len(spec[0])
            xs = [np.stack([np.asarray(sp[i], dtype=np.float64) for sp in spec])[..., np.newaxis]
                  for i in range(n_tow)]
            return xs, _labels(label, len(spec))
        x = np.stack([np.asarray(s, dtype=np.float64) for s in spec])[..., np.newaxis]
        return x, _labels(label, len(spec))
    expanded = np.asarray(spec)[np.newaxis, :, :, np.newaxis]
    gold = None if label is None else np.expand_dims(label, axis=0)
    return expanded, gold
