"""Oracle (test infrastructure): numpy forward pass of RDCNN.res_net.

Restates the Keras graph built in /root/reference/RDCNN.py:176-233 with the
shortcut rule of RDCNN.py:312-335 and the output scaling of RDCNN.py:304-310,
591-597, evaluated with Keras inference semantics (from the Keras docs; TF is
not installed here):
  * Conv2D(padding="same"): NHWC, kernel (kh, kw, Cin, Cout); for even kernels
    pad_before = (k-1)//2, pad_after = k-1-pad_before.
  * BatchNormalization(): epsilon = 1e-3, inference uses moving mean/var.
  * MaxPooling2D / AveragePooling2D: 'valid', strides = pool size.
  * Flatten: row-major over (H, W, C).  Dense: x @ kernel + bias.

PARITY UNPINNED: the reference ships no weights and no recorded network
output, so nothing pins this file against Keras itself.  It
is the spec the HIP forward is compared with, on shared synthetic weights.

Weight dictionary (name -> ndarray), tower t, conv layer i (1-based):
    t{t}/conv{i}/kernel [kh,kw,cin,cout]   t{t}/conv{i}/bias [cout]
    t{t}/bn{i}/{gamma,beta,mean,var} [cout]
    t{t}/sc{i}/kernel [1,1,cin,cout], /bias      (projection, only if channels differ)
    t{t}/scbn{i}/{gamma,beta,mean,var}           (only if shapes differ)
    t{t}/resbn{i}/{gamma,beta,mean,var}          (BN after the Add)
    dense1/kernel [flat,300] dense1/bias ; dense2/kernel [300,K] dense2/bias
"""
import numpy as np

BN_EPS = 1e-3


def sigmoid(x):
    with np.errstate(over='ignore'):
        return 1.0 / (1.0 + np.exp(-x))


def conv2d_same(x, k, b):
    kh, kw, cin, cout = k.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    pb, pr = kh - 1 - pt, kw - 1 - pl
    B, H, W, _ = x.shape
    out = np.empty((B, H, W, cout), dtype=x.dtype)
    k2 = k.reshape(kh * kw * cin, cout)
    for n in range(B):
        xp = np.pad(x[n], ((pt, pb), (pl, pr), (0, 0)))
        win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(0, 1))
        # win: [H, W, cin, kh, kw] -> [H*W, kh*kw*cin]
        col = win.transpose(0, 1, 3, 4, 2).reshape(H * W, kh * kw * cin)
        out[n] = (col @ k2).reshape(H, W, cout)
    return out + b


def batchnorm(x, w, prefix):
    g, be = w[prefix + '/gamma'], w[prefix + '/beta']
    m, v = w[prefix + '/mean'], w[prefix + '/var']
    return (x - m) / np.sqrt(v + x.dtype.type(BN_EPS)) * g + be


def pool2d(x, pool, op):
    ph, pw = pool
    B, H, W, C = x.shape
    Ho, Wo = H // ph, W // pw
    xr = x[:, :Ho * ph, :Wo * pw, :].reshape(B, Ho, ph, Wo, pw, C)
    return xr.max(axis=(2, 4)) if op == 'max' else xr.mean(axis=(2, 4))


def add_shortcut(w, t, i, layer_from, layer_to):           # RDCNN.py:312-335
    sh1, sh2 = list(layer_from.shape[1:]), list(layer_to.shape[1:])
    if sh1 == sh2:
        return layer_from + layer_to
    strides = (int(np.floor(sh1[0] / sh2[0])), int(np.floor(sh1[1] / sh2[1])))
    inter = layer_from
    if sh1[2] != sh2[2]:
        k = w['t%d/sc%d/kernel' % (t, i)]
        inter = inter @ k[0, 0] + w['t%d/sc%d/bias' % (t, i)]
    if sh1[:2] != sh2[:2]:
        inter = pool2d(inter, strides, 'avg')
    inter = batchnorm(inter, w, 't%d/scbn%d' % (t, i))
    return inter + layer_to


def forward(w, cfg, xs, dtype=np.float32, return_logits=False):
    """cfg: dict with input_shapes, kernel_sizes, pool_sizes,
    convolutional_layer_count, feature_expand_frequency, pool_layer_frequency,
    residual_layer_frequencies (list), output_classes, output_range.
    xs: list (one per tower) of [B,H,W,1] arrays.  Returns what
    res_net.predict returns (RDCNN.py:591-597)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    rfreq = cfg['residual_layer_frequencies']
    flats = []
    for t, x in enumerate(xs):
        p1 = np.asarray(x, dtype=dtype)
        p0 = [p1] * len(rfreq)
        for i in range(1, cfg['convolutional_layer_count'] + 1):   # RDCNN.py:185-198
            p1 = conv2d_same(p1, w['t%d/conv%d/kernel' % (t, i)], w['t%d/conv%d/bias' % (t, i)])
            p1 = batchnorm(p1, w, 't%d/bn%d' % (t, i))
            p1 = sigmoid(p1)
            for ri in range(len(rfreq)):
                if i % rfreq[ri] == 0:
                    p1 = add_shortcut(w, t, i, p0[ri], p1)
                    p1 = batchnorm(p1, w, 't%d/resbn%d' % (t, i))
                    p0[ri] = p1
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1 = pool2d(p1, cfg['pool_sizes'][t], 'max')
        flats.append(p1.reshape(p1.shape[0], -1))                  # RDCNN.py:201
    cted = np.concatenate(flats, axis=1) if len(flats) > 1 else flats[0]
    m = sigmoid(cted @ w['dense1/kernel'] + w['dense1/bias'])      # RDCNN.py:212-213
    m = m @ w['dense2/kernel'] + w['dense2/bias']                  # RDCNN.py:214
    if return_logits:
        return m
    if cfg['output_classes'] > 1:                                   # RDCNN.py:218-219
        e = np.exp(m - m.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)
    y = sigmoid(m)                                                  # RDCNN.py:220-221
    lo, hi = cfg['output_range']
    return (y - 0) / 1 * (hi - lo) + lo                            # RDCNN.py:308-310


def head_config(params, name):
    """Constructor arguments of the four classifier heads, restated from
    pitch_classifier.py:12-35, instrumentclassifier.py:17-59,
    velocity_classifier.py:12-35, timing_classifier.py:13-36."""
    base = dict(convolutional_layer_count=params.convolutional_layer_count,
                pool_layer_frequency=params.pool_layer_frequency,
                feature_expand_frequency=params.feature_expand_frequency,
                residual_layer_frequencies=list(params.residual_layer_frequencies))
    if name == 'pitch':
        base.update(input_shapes=[(params.pitch_bands, params.pitch_frames, 1)],
                    kernel_sizes=params.kernel_size_pitch, pool_sizes=params.pool_size_pitch,
                    output_classes=1, output_range=[params.pitch_low, params.pitch_high])
    elif name in ('instrument', 'instrument_focused', 'instrument_focused_const'):
        base.update(input_shapes=[(params.instrument_bands, params.instrument_frames, 1)],
                    kernel_sizes=params.kernel_size_instrument,
                    pool_sizes=params.pool_size_instrument,
                    output_classes=params.instrument_classes, output_range=[0, 128])
    elif name == 'instrument_dual':
        base.update(input_shapes=[(params.instrument_bands, params.instrument_frames, 1)] * 2,
                    kernel_sizes=params.kernel_size_instrument * 2,
                    pool_sizes=params.pool_size_instrument * 2,
                    output_classes=params.instrument_classes, output_range=[0, 128])
    elif name == 'velocity':
        base.update(input_shapes=[(params.bins_velocity, params.pitch_frames, 1)],
                    kernel_sizes=params.kernel_size_velocity,
                    pool_sizes=params.pool_size_velocity,
                    output_classes=1,
                    output_range=[params.velocity_min, params.velocity_max],
                    convolutional_layer_count=params.convolutional_layer_count // 3,
                    pool_layer_frequency=params.pool_layer_frequency // 3,
                    feature_expand_frequency=params.feature_expand_frequency // 3)
    elif name in ('timing', 'timing_start', 'timing_end'):
        base.update(input_shapes=[(params.timing_bands, params.timing_frames, 1)],
                    kernel_sizes=params.kernel_size_timing, pool_sizes=params.pool_size_timing,
                    output_classes=1, output_range=[0, params.timing_frames])
    else:
        raise ValueError('Invalid Variant Selected')
    return base
