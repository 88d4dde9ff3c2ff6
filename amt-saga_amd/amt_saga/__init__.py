"""amt_saga -- host side of the MI355X-native AMT-SAGA hot path.

Python classes that keep the reference's call surface (audio_complete,
res_net, pitch_classifier, InstrumentClassifier, VelocityClassifier,
timming_classifier, Hyperparams) over the C ABI in include/amt_saga.h
(hand-written HIP kernels for gfx950).  PyTorch is used only for device
memory, streams and torch.distributed.  There is no CPU fallback: importing
is cheap, but any compute call raises if the HIP library is not built.
"""
from .hyperparams import Hyperparams, check_shape, list_to_nd_array  # noqa: F401

__all__ = ['Hyperparams', 'check_shape', 'list_to_nd_array']
