"""The four classifier heads: same class names, constructor arguments and
classify() contract as /root/reference/pitch_classifier.py:12-57,
instrumentclassifier.py:11-83, velocity_classifier.py:11-57 and
timing_classifier.py:13-58.  classify(spec, gold=None) predicts; with gold it
trains on the batch, or tests it when test_phase is set (res_net.train / .test).

classify() additionally accepts a device tensor [B, bands, frames] (the batched
loop keeps features in HBM) and then returns a device tensor."""
import torch

from .hyperparams import check_shape, list_to_nd_array
from .rdcnn import res_net


class _Head(res_net):
    _bands = _frames = None

    def _classify(self, spec, gold, test_phase):
        if gold is not None:                               # pitch_classifier.py:48-55: test or train on the batch
            check_shape(spec, self._bands, self._frames)
            expanded, gold_expanded = list_to_nd_array(spec, gold)
            if test_phase:
                return self.test(expanded, gold_expanded, use_predict=True)
            return self.train(expanded, gold_expanded)
        if isinstance(spec, torch.Tensor):
            if tuple(spec.shape[1:3]) != (self._bands, self._frames):
                raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format(
                    (self._bands, self._frames), tuple(spec.shape[1:3])))
            return self.predict_device([spec])
        if isinstance(spec, (list, tuple)) and isinstance(spec[0], torch.Tensor) \
                and spec[0].dim() == 3:
            return self.predict_device(list(spec))
        check_shape(spec, self._bands, self._frames)
        expanded, _ = list_to_nd_array(spec, None)
        return self.predict(expanded)


class pitch_classifier(_Head):
    def __init__(self, params, checkpoint_prefix='checkpoint_pitch',
                 metrics_prefix='metrics_pitch', weight_seed=101, calibrated=True):
        super().__init__(input_shapes=[(params.pitch_bands, params.pitch_frames, 1)],
                         kernel_sizes=params.kernel_size_pitch,
                         pool_sizes=params.pool_size_pitch,
                         output_classes=1,
                         output_range=[params.pitch_low, params.pitch_high],
                         batch_size=params.batch_size,
                         convolutional_layer_count=params.convolutional_layer_count,
                         pool_layer_frequency=params.pool_layer_frequency,
                         feature_expand_frequency=params.feature_expand_frequency,
                         residual_layer_frequencies=params.residual_layer_frequencies,
                         checkpoint_dir=params.checkpoint_dir,
                         checkpoint_frequency=params.checkpoint_frequency,
                         checkpoint_prefix=checkpoint_prefix,
                         metrics_prefix=metrics_prefix, metrics=[],
                         logging_parent='AMT-SAGA', weight_seed=weight_seed, calibrated=calibrated)
        self.params = params
        self._bands, self._frames = params.pitch_bands, params.pitch_frames

    def classify(self, spec, pitch_gold=None, test_phase=False):
        return self._classify(spec, pitch_gold, test_phase)


class InstrumentClassifier(_Head):
    INSTRUMENT = 'instrument'
    INSTRUMENT_FOCUSED = 'instrument_focused'
    INSTRUMENT_FOCUSED_CONST = 'instrument_focused_const'
    INSTRUMENT_DUAL = 'instrument_dual'

    def __init__(self, params, variant, prefix=None, weight_seed=102, calibrated=True):
        if variant in (InstrumentClassifier.INSTRUMENT, InstrumentClassifier.INSTRUMENT_FOCUSED,
                       InstrumentClassifier.INSTRUMENT_FOCUSED_CONST):
            input_shape = [(params.instrument_bands, params.instrument_frames, 1)]
            kernel_size = params.kernel_size_instrument
            pool_size = params.pool_size_instrument
        elif variant == InstrumentClassifier.INSTRUMENT_DUAL:
            input_shape = [(params.instrument_bands, params.instrument_frames, 1)] * 2
            kernel_size = params.kernel_size_instrument * 2
            pool_size = params.pool_size_instrument * 2
        else:
            raise ValueError('Invalid Variant Selected')
        name = variant if prefix is None else prefix
        super().__init__(input_shapes=input_shape, kernel_sizes=kernel_size, pool_sizes=pool_size,
                         output_classes=params.instrument_classes,
                         batch_size=params.batch_size,
                         convolutional_layer_count=params.convolutional_layer_count,
                         pool_layer_frequency=params.pool_layer_frequency,
                         feature_expand_frequency=params.feature_expand_frequency,
                         residual_layer_frequencies=params.residual_layer_frequencies,
                         checkpoint_dir=params.checkpoint_dir,
                         checkpoint_frequency=params.checkpoint_frequency,
                         checkpoint_prefix='checkpoint_' + name,
                         metrics_prefix='metrics_' + name, metrics=[],
                         logging_parent='AMT-SAGA', weight_seed=weight_seed, calibrated=calibrated)
        self.params = params
        self.variant = variant
        self._bands, self._frames = params.instrument_bands, params.instrument_frames

    def classify(self, spec, instrument_gold=None, test_phase=False):
        return self._classify(spec, instrument_gold, test_phase)


class VelocityClassifier(_Head):
    def __init__(self, params, checkpoint_prefix='checkpoint_velocity',
                 metrics_prefix='metrics_velocity', weight_seed=103, calibrated=True):
        super().__init__(input_shapes=[(params.bins_velocity, params.pitch_frames, 1)],
                         kernel_sizes=params.kernel_size_velocity,
                         pool_sizes=params.pool_size_velocity,
                         output_classes=1,
                         output_range=[params.velocity_min, params.velocity_max],
                         batch_size=params.batch_size,
                         convolutional_layer_count=params.convolutional_layer_count // 3,
                         pool_layer_frequency=params.pool_layer_frequency // 3,
                         feature_expand_frequency=params.feature_expand_frequency // 3,
                         residual_layer_frequencies=params.residual_layer_frequencies,
                         checkpoint_dir=params.checkpoint_dir,
                         checkpoint_frequency=params.checkpoint_frequency,
                         checkpoint_prefix=checkpoint_prefix,
                         metrics_prefix=metrics_prefix, metrics=[],
                         logging_parent='AMT-SAGA', weight_seed=weight_seed, calibrated=calibrated)
        self.params = params
        self._bands, self._frames = params.bins_velocity, params.pitch_frames

    def classify(self, spec, velocity_gold=None, test_phase=False):
        return self._classify(spec, velocity_gold, test_phase)


class timming_classifier(_Head):
    def __init__(self, params, checkpoint_prefix='checkpoint_timing',
                 metrics_prefix='metrics_timing', weight_seed=104, calibrated=True):
        super().__init__(input_shapes=[(params.timing_bands, params.timing_frames, 1)],
                         kernel_sizes=params.kernel_size_timing,
                         pool_sizes=params.pool_size_timing,
                         output_classes=1,
                         output_range=[0, params.timing_frames],
                         batch_size=params.batch_size,
                         convolutional_layer_count=params.convolutional_layer_count,
                         pool_layer_frequency=params.pool_layer_frequency,
                         feature_expand_frequency=params.feature_expand_frequency,
                         residual_layer_frequencies=params.residual_layer_frequencies,
                         checkpoint_dir=params.checkpoint_dir,
                         checkpoint_frequency=params.checkpoint_frequency,
                         checkpoint_prefix=checkpoint_prefix,
                         metrics_prefix=metrics_prefix, metrics=[],
                         logging_parent='AMT-SAGA', weight_seed=weight_seed, calibrated=calibrated)
        self.params = params
        self._bands, self._frames = params.timing_bands, params.timing_frames

    def classify(self, spec, timing_gold=None, test_phase=False):
        return self._classify(spec, timing_gold, test_phase)
