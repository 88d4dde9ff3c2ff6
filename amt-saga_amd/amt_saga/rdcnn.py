"""res_net: the residual deep CNN of /root/reference/RDCNN.py, inference only.

Keeps the reference constructor's argument names (RDCNN.py:42-63) and
``predict(x)`` (RDCNN.py:591-597); the graph is the one RDCNN.py:176-233
builds.  The forward pass runs in the HIP library (amt_rdcnn_forward: fp32
MFMA implicit-GEMM convolutions, fused BN/sigmoid/shortcut epilogues);
train / test / fit_generator (RDCNN.py:503-589) run amt_trainer_step (training-mode BN, backward,
Adagrad, MSE / sparse-CCE); report / plot / metrics bookkeeping stay out of scope.

Weights.  The reference ships none; ``init_weights(seed)`` draws synthetic
ones (Keras initialisers for kernels; BN moving statistics and biases are
randomised so that they matter).  Canonical order of the blob handed to the C
ABI (``pack_weights``), per tower t, conv layer i = 1..L in graph order:
    conv kernel [kh,kw,cin,cout], conv bias, bn gamma, beta, moving mean, var
    if i is a shortcut layer:
        if shapes differ: [1x1 kernel [cin_from,cout], bias if channels differ],
                          shortcut bn gamma, beta, mean, var
        post-add bn gamma, beta, mean, var
then dense1 kernel [flat,300], bias, dense2 kernel [300,K], bias.
Names follow oracle/rdcnn.py ("t0/conv1/kernel", ...); ``load_weights`` reads
an .npz with those names.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import torch

from . import _lib
from .device import empty, ptr, require_gpu, stream_ptr, to_dev


# convolution arithmetic (all f32-equivalent): 3 (default) = split-fp16 (3 f16 MFMAs per product block) with the
# 32 -> 32 (4 x 16) layers on the timing heads' large images in the FFT domain (amt_fftconv.hip), 2 = split-fp16
# everywhere, 1 = split-bf16 (6 bf16 MFMAs), 0 = f32 MFMA
DEFAULT_MODE = int(os.environ.get('AMT_CONV_MODE', '3'))

# BatchNorm statistics / last-Dense scaling that make the seeded synthetic heads input-sensitive
# (tests/golden/gen_synthetic_calibration.py); keys "<topology signature>/<tensor name>"
_CALIBRATION_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'synthetic_heads.npz')
_calibration = None


def topology_signature(cfg, seed):
    """Identifies (graph, seed): the calibration table is valid for exactly that weight draw."""
    blob = json.dumps({k: cfg[k] for k in sorted(cfg)}, sort_keys=True, default=list)
    return '%s-s%d' % (hashlib.sha1(blob.encode()).hexdigest()[:10], int(seed))


def _calibration_table():
    global _calibration
    if _calibration is None:
        if os.path.exists(_CALIBRATION_FILE):
            with np.load(_CALIBRATION_FILE) as z:
                _calibration = {k: z[k] for k in z.files}
        else:
            _calibration = {}
    return _calibration


def _to_list(single):
    """RDCNN.py:288-302."""
    if single is None:
        return []
    try:
        len(single)
        return list(single)
    except TypeError:
        return [] if single < 1 else [single]


class res_net:
    def __init__(self,
                 input_shapes=[(20, 2049, 1,), (20, 256, 1,)],
                 output_classes=128,
                 output_range=[0, 128],
                 batch_size=1,
                 kernel_sizes=[(3, 32), (3, 8)],
                 pool_sizes=[(2, 5), (2, 5)],
                 convolutional_layer_count=15,
                 feature_expand_frequency=6,
                 pool_layer_frequency=6,
                 residual_layer_frequencies=2,
                 metrics=None,
                 checkpoint_dir=None, checkpoint_prefix='checkpoint',
                 checkpoint_frequency=5000,
                 metrics_prefix='metrics_logs',
                 weights_load_checkpoint_filename=None,
                 starting_checkpoint_index=1,
                 logging_parent=None,
                 weight_seed=1234, calibrated=True):
        self.batch_size = batch_size          # the HIP forward accepts any B
        self.out_func_min = 0
        self.out_func_max = 1
        self.out_func_factor = 1
        self.output_classes = int(output_classes)
        rf = _to_list(residual_layer_frequencies)
        if len(rf) > 1:
            raise ValueError('only one residual frequency is supported by the HIP forward')
        if output_classes > 1:
            pass
        elif output_classes == 1:
            self.out_val_min = output_range[0]
            self.out_val_max = output_range[1]
            self.out_val_factor = output_range[1] - output_range[0]
        else:
            raise ValueError('Invalid output size {}'.format(str(output_classes)))
        if not 1 <= len(input_shapes) <= 2:
            raise ValueError('1 or 2 input towers are supported')
        self.cfg = dict(
            input_shapes=[tuple(s) for s in input_shapes],
            kernel_sizes=[tuple(k) for k in kernel_sizes][:len(input_shapes)],
            pool_sizes=[tuple(p) for p in pool_sizes][:len(input_shapes)],
            convolutional_layer_count=int(convolutional_layer_count),
            feature_expand_frequency=int(feature_expand_frequency or 0),
            pool_layer_frequency=int(pool_layer_frequency or 0),
            residual_layer_frequencies=rf,
            output_classes=int(output_classes),
            output_range=list(output_range))
        self.checkpoint_dir = checkpoint_dir
        self.checkpoint_prefix = checkpoint_prefix
        self.checkpoint_frequency = checkpoint_frequency
        self.metrics_prefix = metrics_prefix
        self.current_batch = starting_checkpoint_index if starting_checkpoint_index is not None else 1
        self.metrics_train = []
        self.metrics_test = []
        self._net = None
        self._ws = None
        self._weights = None
        self.mode = DEFAULT_MODE
        self.layout = self._walk()
        if weights_load_checkpoint_filename is not None:
            self.load_weights(weights_load_checkpoint_filename)
        else:
            self.set_weights(self.init_weights(weight_seed, calibrated))

    # ---- topology (RDCNN.py:176-233) ---------------------------------------------
    def _walk(self):
        """[(name, shape)] in canonical blob order + flatten size."""
        c = self.cfg
        r = c['residual_layer_frequencies'][0] if c['residual_layer_frequencies'] else 0
        out = []
        flat = 0
        for t, (H, W, C0) in enumerate(c['input_shapes']):
            if C0 != 1:
                raise ValueError('input towers must have one channel')
            kh, kw = c['kernel_sizes'][t]
            ph, pw = c['pool_sizes'][t]
            Cc, fo = 1, 32
            p0 = (H, W, 1)
            for i in range(1, c['convolutional_layer_count'] + 1):
                out.append(('t%d/conv%d/kernel' % (t, i), (kh, kw, Cc, fo)))
                out.append(('t%d/conv%d/bias' % (t, i), (fo,)))
                for k in ('gamma', 'beta', 'mean', 'var'):
                    out.append(('t%d/bn%d/%s' % (t, i, k), (fo,)))
                Cc = fo
                if r and i % r == 0:
                    if p0 != (H, W, Cc):
                        if p0[2] != Cc:
                            out.append(('t%d/sc%d/kernel' % (t, i), (1, 1, p0[2], Cc)))
                            out.append(('t%d/sc%d/bias' % (t, i), (Cc,)))
                        for k in ('gamma', 'beta', 'mean', 'var'):
                            out.append(('t%d/scbn%d/%s' % (t, i, k), (Cc,)))
                    for k in ('gamma', 'beta', 'mean', 'var'):
                        out.append(('t%d/resbn%d/%s' % (t, i, k), (Cc,)))
                    p0 = (H, W, Cc)
                if c['pool_layer_frequency'] and i % c['pool_layer_frequency'] == 0:
                    H, W = H // ph, W // pw
                if c['feature_expand_frequency'] and i % c['feature_expand_frequency'] == 0:
                    fo *= 2
            flat += H * W * Cc
        out.append(('dense1/kernel', (flat, 300)))
        out.append(('dense1/bias', (300,)))
        out.append(('dense2/kernel', (300, c['output_classes'])))
        out.append(('dense2/bias', (c['output_classes'],)))
        self.flat = flat
        return out

    def init_weights(self, seed=1234, calibrated=True):
        """Synthetic weights (the reference ships none): glorot-uniform kernels (Keras default), small
        random biases, random BN parameters.  When the calibration table holds an entry for this
        (topology, seed) -- the heads' default seeds do -- the BN parameters and the last Dense are
        replaced by the calibrated ones: BN moving statistics equal to the statistics of the layer's
        own input on the loop's features, as a trained model has them, so that the head's output
        moves with its input (tests/golden/gen_synthetic_calibration.py)."""
        rng = np.random.default_rng(seed)
        w = {}
        for name, shape in self.layout:
            kind = name.rsplit('/', 1)[1]
            if kind == 'kernel':
                if len(shape) == 4:
                    fan_in = shape[0] * shape[1] * shape[2]
                    fan_out = shape[0] * shape[1] * shape[3]
                else:
                    fan_in, fan_out = shape
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-lim, lim, shape)
            elif kind == 'bias':
                a = rng.normal(0, 0.05, shape)
            elif kind == 'gamma':
                a = rng.uniform(2.5, 4.5, shape)
            elif kind in ('beta', 'mean'):
                a = rng.normal(0, 0.2, shape)
            else:
                a = rng.uniform(0.5, 1.5, shape)
            w[name] = a.astype(np.float32)
        self.calibrated = False
        if calibrated:
            tab, sig = _calibration_table(), topology_signature(self.cfg, seed) + '/'
            for name, shape in self.layout:
                a = tab.get(sig + name)
                if a is not None:
                    if tuple(a.shape) != tuple(shape):
                        raise ValueError('Invalid Input shape. Expected: {} . Got: {} ({})'.format(
                            shape, a.shape, name))
                    w[name] = np.asarray(a, np.float32)
                    self.calibrated = True
        return w

    def pack_weights(self, w):
        parts = []
        for name, shape in self.layout:
            a = np.asarray(w[name], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError('Invalid Input shape. Expected: {} . Got: {} ({})'.format(
                    shape, a.shape, name))
            parts.append(a.reshape(-1))
        return np.ascontiguousarray(np.concatenate(parts))

    def _desc(self):
        c = self.cfg
        d = _lib.RdcnnDesc()
        d.n_towers = len(c['input_shapes'])
        for t in range(d.n_towers):
            d.in_h[t], d.in_w[t] = c['input_shapes'][t][0], c['input_shapes'][t][1]
            d.kh[t], d.kw[t] = c['kernel_sizes'][t]
            d.pool_h[t], d.pool_w[t] = c['pool_sizes'][t]
        d.conv_layers = c['convolutional_layer_count']
        d.feature_expand_frequency = c['feature_expand_frequency']
        d.pool_layer_frequency = c['pool_layer_frequency']
        d.residual_frequency = c['residual_layer_frequencies'][0] if c['residual_layer_frequencies'] else 0
        d.dense_units = 300
        d.output_classes = c['output_classes']
        if c['output_classes'] == 1:
            d.out_lo, d.out_hi = float(c['output_range'][0]), float(c['output_range'][1])
        else:
            d.out_lo, d.out_hi = 0.0, 1.0
        return d

    def set_weights(self, w):
        """Upload a weight dict (see module docstring) to the device."""
        self._weights = w
        self._blob = self.pack_weights(w)
        self._release()

    def _ensure(self):
        self._sync_from_trainer()
        if self._net is not None:
            return
        lib = _lib.load()
        require_gpu()
        d = self._desc()
        n = lib.amt_rdcnn_param_count(C.byref(d))
        if n != self._blob.size:
            raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format(n, self._blob.size))
        h = C.c_void_p()
        _lib.check(lib.amt_rdcnn_create(C.byref(h), C.byref(d),
                                        self._blob.ctypes.data_as(C.c_void_p), self._blob.size))
        self._net = h
        self._lib = lib
        _lib.check(lib.amt_rdcnn_set_mode(h, int(getattr(self, 'mode', DEFAULT_MODE))))

    def _release(self):
        if getattr(self, '_net', None) is not None:
            self._lib.amt_rdcnn_destroy(self._net)
        self._net = None
        self._ws = None
        if getattr(self, '_trainer', None) is not None:
            self._tlib.amt_trainer_destroy(self._trainer)
        self._trainer = None
        self._trained = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_weights(self, filename):                       # RDCNN.py:778-782
        """The reference's checkpoints are Keras ``save_weights`` HDF5 files ('.h5', RDCNN.py:490-494): read
        by the pure-Python importer (keras_io / hdf5).  '.npz' = this build's own save_weights()."""
        with open(filename, 'rb') as fh:
            magic = fh.read(8)
        if magic == b'\x89HDF\r\n\x1a\n':
            from .keras_io import load_keras_weights
            self.set_weights(load_keras_weights(filename, self.layout))
            return
        with np.load(filename) as z:
            self.set_weights({k: z[k] for k in z.files})

    def save_weights(self, filename):
        """'.h5' / '.hdf5': the Keras ``save_weights`` file of this graph (RDCNN.py:490-494; loadable by the
        reference's ``load_weights``); otherwise this build's '.npz'."""
        self._sync_from_trainer()
        if str(filename).endswith(('.h5', '.hdf5')):
            from .keras_io import save_keras_weights
            save_keras_weights(filename, self.weights, self.cfg)
        else:
            np.savez(filename, **self.weights)

    @property
    def flops_per_window(self):
        self._ensure()
        return float(self._lib.amt_rdcnn_flops_per_window(self._net))

    def set_mode(self, mode):
        """0 = f32 MFMA convolutions, 1 = split-bf16 (6 bf16 MFMAs per product block), 2 = split-fp16 (3 f16 MFMAs per
        product block), 3 = split-fp16 + the FFT-domain form of the 32 -> 32 (4 x 16) layers on images of at most 20 x 561
        (nets without such layers run exactly as in mode 2); all f32-equivalent."""
        self.mode = int(mode)
        if self._net is not None:
            _lib.check(self._lib.amt_rdcnn_set_mode(self._net, self.mode))

    # ---- measurement hook ------------------------------------------------------------
    def profile(self, enable=True):
        self._ensure()
        _lib.check(self._lib.amt_rdcnn_profile(self._net, int(bool(enable))))

    def profile_read(self, reset=True):
        """[{tower, layer, kh, kw, cin, cout, H, W, ms, windows, flops_per_window}] per conv layer."""
        self._ensure()
        n = C.c_int32(0)
        _lib.check(self._lib.amt_rdcnn_profile_read(self._net, None, None, None, None, 0, C.byref(n), 0))
        cap = n.value
        desc = np.zeros((cap, 8), np.int32)
        ms = np.zeros(cap); win = np.zeros(cap); fl = np.zeros(cap)
        _lib.check(self._lib.amt_rdcnn_profile_read(
            self._net, desc.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p),
            win.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p), cap, C.byref(n), int(bool(reset))))
        keys = ('tower', 'layer', 'kh', 'kw', 'cin', 'cout', 'H', 'W')
        return [dict(zip(keys, map(int, desc[r])), ms=float(ms[r]), windows=float(win[r]),
                     flops_per_window=float(fl[r])) for r in range(cap)]

    # ---- forward ---------------------------------------------------------------
    def predict_device(self, xs, return_logits=False):
        """xs: list (one per tower) of device tensors [B, H, W] f32 (NHWC, C=1).
        Returns a device tensor [B, K] (and the logits if asked)."""
        self._ensure()
        B = xs[0].shape[0]
        for t, x in enumerate(xs):
            H, W, _ = self.cfg['input_shapes'][t]
            if tuple(x.shape[1:3]) != (H, W) or x.shape[0] != B:
                raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format(
                    (H, W), tuple(x.shape[1:3])))
        need = self._lib.amt_rdcnn_workspace_bytes(self._net, B)
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = empty(((need + 3) // 4,))
        K = self.cfg['output_classes']
        y = empty((B, K))
        lg = empty((B, K)) if return_logits else None
        arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        _lib.check(self._lib.amt_rdcnn_forward(self._net, arr, B, ptr(y), ptr(lg), ptr(self._ws),
                                               self._ws.numel() * 4, stream_ptr()))
        return (y, lg) if return_logits else y

    def predict(self, x):
        """RDCNN.py:591-597: x is (B,H,W,1) (or a list of such, one per tower);
        returns the scaled regression value (B,1) or softmax probabilities (B,K)."""
        xs = x if isinstance(x, (list, tuple)) else [x]
        dx = []
        for a in xs:
            a = a if isinstance(a, torch.Tensor) else np.asarray(a)
            if a.ndim == 4:
                a = a[..., 0]
            dx.append(to_dev(a))
        return self.predict_device(dx).cpu().numpy()

    def _scale_output_to_activation(self, x):               # RDCNN.py:304-306
        return ((x - self.out_val_min) / self.out_val_factor) * self.out_func_factor + self.out_func_min

    def _scale_activation_to_output(self, x):               # RDCNN.py:308-310
        return ((x - self.out_func_min) / self.out_func_factor) * self.out_val_factor + self.out_val_min

    @property
    def weights(self):
        """Canonical weight dict.  After train() the live weights are on the device; reading this pulls them back
        first, so the dict is never stale (bench / smoke / save read it directly)."""
        self._sync_from_trainer()
        return self._weights

    @weights.setter
    def weights(self, w):
        self._weights = w

    # Keras' Adagrad defaults changed TOGETHER between versions, and the reference (``keras.optimizers.Adagrad()``,
    # RDCNN.py:246-253) pins none: Keras 2.2 / tf.keras 1.13: lr 0.01, accumulator 0, epsilon K.epsilon() = 1e-7;
    # tf.keras >= 1.14 (optimizer_v2): lr 0.001, initial accumulator 0.1, epsilon 1e-7.  ``keras_optimizer_version``
    # selects one consistent triple (default: the reference's era, 'keras-2.2'); the individual attributes
    # ``learning_rate`` / ``adagrad_epsilon`` / ``adagrad_initial_accumulator`` still override it.
    ADAGRAD_DEFAULTS = {'keras-2.2': (0.01, 1e-7, 0.0), 'tf.keras-1.14': (0.001, 1e-7, 0.1)}
    keras_optimizer_version = 'keras-2.2'

    # ---- training side (RDCNN.py:503-589): amt_trainer_step on the device ----------------------------
    def _trainer_handle(self):
        """The device-side trainer, created from the current weights on first use.  From then on it owns
        the live weights; predict() pulls them back before its next forward (``_sync_from_trainer``).
        Optimiser = Adagrad as RDCNN.py:246-253 compiles it; ``keras_optimizer_version`` picks the (lr, epsilon,
        initial accumulator) triple of one Keras generation (see ADAGRAD_DEFAULTS); ``learning_rate``,
        ``adagrad_epsilon`` and ``adagrad_initial_accumulator`` override single values; all before the first train()."""
        if getattr(self, '_trainer', None) is None:
            lib = _lib.load()
            require_gpu()
            d = self._desc()
            h = C.c_void_p()
            if self.keras_optimizer_version not in self.ADAGRAD_DEFAULTS:
                raise ValueError('Requested attribute does not exist')
            lr0, eps0, acc0 = self.ADAGRAD_DEFAULTS[self.keras_optimizer_version]
            _lib.check(lib.amt_trainer_create(C.byref(h), C.byref(d), self._blob.ctypes.data_as(C.c_void_p),
                                              self._blob.size, float(getattr(self, 'learning_rate', lr0)),
                                              float(getattr(self, 'adagrad_epsilon', eps0)),
                                              float(getattr(self, 'adagrad_initial_accumulator', acc0))))
            self._trainer, self._tlib, self._trained = h, lib, False
        return self._trainer

    def _sync_from_trainer(self):
        if getattr(self, '_trainer', None) is not None and self._trained:
            blob = np.empty(self._blob.size, np.float32)
            _lib.check(self._tlib.amt_trainer_get_weights(self._trainer, blob.ctypes.data_as(C.c_void_p), blob.size))
            w, off = {}, 0
            for name, shape in self.layout:
                n = int(np.prod(shape))
                w[name] = blob[off:off + n].reshape(shape).copy()
                off += n
            self._weights, self._blob = w, blob
            if self._net is not None:
                self._lib.amt_rdcnn_destroy(self._net)
            self._net = None
            self._trained = False

    def gradients(self):
        """Gradients of the last train() call as a weight-shaped dict (zeros for BN moving statistics)."""
        h = self._trainer_handle()                      # creates the trainer (and self._tlib) if train() has not run yet
        blob = np.empty(self._blob.size, np.float32)
        _lib.check(self._tlib.amt_trainer_get_grads(h, blob.ctypes.data_as(C.c_void_p), blob.size))
        out, off = {}, 0
        for name, shape in self.layout:
            n = int(np.prod(shape))
            out[name] = blob[off:off + n].reshape(shape).copy()
            off += n
        return out

    def _batch(self, x, y, update):
        xs = x if isinstance(x, (list, tuple)) else [x]
        dx = []
        for a in xs:
            a = a if isinstance(a, torch.Tensor) else np.asarray(a)
            if a.ndim == 4:
                a = a[..., 0]
            dx.append(to_dev(a))
        B = dx[0].shape[0]
        yv = np.asarray(y.cpu().numpy() if isinstance(y, torch.Tensor) else y, dtype=np.float64).reshape(-1)
        if yv.shape[0] != B:
            raise ValueError('Invalid Input shape. Expected: {} . Got: {}'.format((B,), yv.shape))
        if self.output_classes == 1:
            yv = self._scale_output_to_activation(yv)               # RDCNN.py:513-514, :569-570
        h = self._trainer_handle()
        K = self.cfg['output_classes']
        pred = empty((B, K))
        loss = C.c_float(0.0)
        arr = (C.c_void_p * len(dx))(*[t.data_ptr() for t in dx])
        y_dev = to_dev(yv.astype(np.float32))           # kept alive across the (asynchronous) step: no temporary
        _lib.check(self._tlib.amt_trainer_step(h, arr, ptr(y_dev), B, int(update),
                                               C.byref(loss), ptr(pred), stream_ptr()))
        p = pred.cpu().numpy()
        del y_dev
        return float(loss.value), (self._scale_activation_to_output(p) if self.output_classes == 1 else p)

    def train(self, x, y):
        """RDCNN.py:503-526: one ``train_on_batch`` (BatchNormalization in training mode, backward, Adagrad;
        targets scaled to the activation range for a single output).  Returns the prediction of the batch in
        output scale ((B,1)) or the class probabilities ((B,K)); the loss is appended to ``metrics_train``."""
        loss, pred = self._batch(x, y, True)
        self._trained = True
        self.metrics_train.append([loss])
        if self.checkpoint_dir and (self.current_batch % self.checkpoint_frequency == 0):
            self.save_checkpoint()
        self.current_batch += 1
        return pred

    def test(self, x, y, use_predict=False):
        """RDCNN.py:559-589: ``test_on_batch`` (inference-mode forward + loss, no update); the loss is appended
        to ``metrics_test``.  ``use_predict`` only changes the reference's bookkeeping, not the numbers."""
        loss, pred = self._batch(x, y, False)
        self.metrics_test.append([loss])
        return pred

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, **kwargs):   # RDCNN.py:536-556
        """Keras fit_generator over (x, y) batches: train() on each."""
        for _ in range(int(epochs)):
            for step, (xb, yb) in enumerate(generator):
                self.train(xb, yb)
                if steps_per_epoch is not None and step + 1 >= steps_per_epoch:
                    break

    def save_checkpoint(self):                              # RDCNN.py:467-501
        """Weights of the current batch index as '<dir>/<prefix>_<batch>.h5', the file name and format the
        reference writes (RDCNN.py:490-494; its metrics bookkeeping is out of scope)."""
        import os as _os
        _os.makedirs(self.checkpoint_dir, exist_ok=True)
        path = _os.path.join(self.checkpoint_dir, '%s_%d.h5' % (self.checkpoint_prefix, self.current_batch))
        self.save_weights(path)
        return path
