"""Oracle (test infrastructure): numpy restatement of one Keras ``train_on_batch`` of RDCNN.res_net --
forward in training mode, hand-written backward (no autograd), Adagrad update.

Follows /root/reference/RDCNN.py: compile() with ``keras.optimizers.Adagrad()`` and
``mean_squared_error`` (one output, targets scaled to the activation range, :304-306, :513-514) or
``sparse_categorical_crossentropy`` (:245-254); ``train()`` = ``model.train_on_batch`` (:503-526).
Keras / TensorFlow are not available here, so their training semantics are restated from the Keras 2.2
documentation and source as remembered -- PARITY UNPINNED against Keras itself:
  * BatchNormalization(training=True): normalise with the batch mean and the biased batch variance over
    (N, H, W); epsilon 1e-3; moving statistics m <- 0.99 m + 0.01 batch, where the variance entering the moving
    average is the UNBIASED one, var * M / (M - 1) (TensorFlow's fused batch norm; Keras 2.2's own path uses
    M / (M - (1 + epsilon)), 1e-3 / M apart), M = samples per channel;
  * Adagrad: lr 0.01, a += g^2, p -= lr g / (sqrt(a) + 1e-7); the accumulators start at 0 in Keras 2.2 /
    tf.keras 1.13 and at 0.1 in tf.keras >= 1.14 -- a parameter here, default 0.1;
  * mean_squared_error: mean over the output axis, then over the batch; sparse_categorical_crossentropy
    on softmax probabilities, mean over the batch (the probability clipping at 1e-7 is not reproduced).
What IS pinned: the backward pass below is checked against PyTorch autograd on the same graph in float64
(tests/test_oracle_train.py), so the gradients are the gradients of the stated forward.
"""
import numpy as np

from . import rdcnn as orc

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


# ---- layers: forward returns (out, cache), backward returns (dx, {param grads}) ---------------------
def _im2col(x, kh, kw):
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))      # [B,H,W,C,kh,kw]
    B, H, W, C = x.shape
    return win.transpose(0, 1, 2, 4, 5, 3).reshape(B * H * W, kh * kw * C)


def conv_fwd(x, k, b):
    kh, kw, cin, cout = k.shape
    col = _im2col(x, kh, kw)
    out = (col @ k.reshape(-1, cout) + b).reshape(x.shape[0], x.shape[1], x.shape[2], cout)
    return out, (x.shape, col, k)


def conv_bwd(dout, cache):
    xshape, col, k = cache
    kh, kw, cin, cout = k.shape
    B, H, W, _ = xshape
    d2 = dout.reshape(-1, cout)
    dk = (col.T @ d2).reshape(k.shape)
    db = d2.sum(axis=0)
    dcol = (d2 @ k.reshape(-1, cout).T).reshape(B, H, W, kh, kw, cin)
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    dxp = np.zeros((B, H + kh - 1, W + kw - 1, cin), dout.dtype)
    for dy in range(kh):
        for dx in range(kw):
            dxp[:, dy:dy + H, dx:dx + W, :] += dcol[:, :, :, dy, dx, :]
    return dxp[:, pt:pt + H, pl:pl + W, :], dk, db


def bn_fwd(z, gamma, beta):
    mu = z.mean(axis=(0, 1, 2))
    var = ((z - mu) ** 2).mean(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + z.dtype.type(BN_EPS))
    zh = (z - mu) * inv
    return zh * gamma + beta, (zh, inv, gamma), mu, var


def bn_bwd(dy, cache):
    zh, inv, gamma = cache
    dgamma = (dy * zh).sum(axis=(0, 1, 2))
    dbeta = dy.sum(axis=(0, 1, 2))
    n = zh.shape[0] * zh.shape[1] * zh.shape[2]
    dz = (gamma * inv) * (dy - dbeta / n - zh * (dgamma / n))
    return dz, dgamma, dbeta


def maxpool_fwd(x, pool):
    ph, pw = pool
    B, H, W, C = x.shape
    Ho, Wo = H // ph, W // pw
    xr = x[:, :Ho * ph, :Wo * pw, :].reshape(B, Ho, ph, Wo, pw, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, Ho, Wo, C, ph * pw)
    idx = xr.argmax(axis=-1)                                  # first maximum in (dy, dx) row-major order
    return np.take_along_axis(xr, idx[..., None], -1)[..., 0], (x.shape, pool, idx)


def maxpool_bwd(dout, cache):
    xshape, (ph, pw), idx = cache
    B, H, W, C = xshape
    Ho, Wo = H // ph, W // pw
    d = np.zeros((B, Ho, Wo, C, ph * pw), dout.dtype)
    np.put_along_axis(d, idx[..., None], dout[..., None], -1)
    dx = np.zeros(xshape, dout.dtype)
    dx[:, :Ho * ph, :Wo * pw, :] = d.reshape(B, Ho, Wo, C, ph, pw).transpose(0, 1, 4, 2, 5, 3).reshape(B, Ho * ph, Wo * pw, C)
    return dx


def avgpool_fwd(x, pool):
    return orc.pool2d(x, pool, 'avg'), (x.shape, pool)


def avgpool_bwd(dout, cache):
    xshape, (ph, pw) = cache
    B, H, W, C = xshape
    Ho, Wo = H // ph, W // pw
    dx = np.zeros(xshape, dout.dtype)
    dx[:, :Ho * ph, :Wo * pw, :] = np.repeat(np.repeat(dout, ph, axis=1), pw, axis=2) / (ph * pw)
    return dx


# ---- the graph of RDCNN.py:176-233 in training mode ---------------------------------------------------
def forward_backward(w, cfg, xs, y, dtype=np.float64):
    """Loss, prediction (activation scale / probabilities), gradients of every trainable tensor and the batch
    statistics of every BN layer.  y: [B] class indices (K > 1) or [B] targets in activation scale (K = 1)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    r = cfg['residual_layer_frequencies'][0] if cfg['residual_layer_frequencies'] else 0
    tapes, flats, stats = [], [], {}
    counts = forward_backward.last_counts = {}          # samples per channel of every BN layer (Bessel's correction)

    def bn(z, prefix, tape):
        out, cache, mu, var = bn_fwd(z, w[prefix + '/gamma'], w[prefix + '/beta'])
        stats[prefix] = (mu, var)
        counts[prefix] = z.size // z.shape[-1]
        tape.append(('bn', prefix, cache))
        return out

    for t, x in enumerate(xs):
        tape = []
        p1 = np.asarray(x, dtype=dtype)
        p0_id = 0                                  # tape position whose output is the shortcut source (0 = input)
        acts = [p1]                                # acts[i] = output after tape entry i-1
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            z, c = conv_fwd(p1, w['t%d/conv%d/kernel' % (t, i)], w['t%d/conv%d/bias' % (t, i)])
            tape.append(('conv', 't%d/conv%d' % (t, i), c))
            z = bn(z, 't%d/bn%d' % (t, i), tape)
            p1 = orc.sigmoid(z)
            tape.append(('sigmoid', None, p1))
            if r and i % r == 0:
                src = acts[p0_id]
                sc_tape = []
                a = src
                if a.shape != p1.shape:
                    if a.shape[-1] != p1.shape[-1]:
                        a, c = conv_fwd(a, w['t%d/sc%d/kernel' % (t, i)], w['t%d/sc%d/bias' % (t, i)])
                        sc_tape.append(('conv', 't%d/sc%d' % (t, i), c))
                    if a.shape[1:3] != p1.shape[1:3]:
                        a, c = avgpool_fwd(a, (a.shape[1] // p1.shape[1], a.shape[2] // p1.shape[2]))
                        sc_tape.append(('avgpool', None, c))
                    a = bn(a, 't%d/scbn%d' % (t, i), sc_tape)
                tape.append(('add', p0_id, sc_tape))
                p1 = bn(a + p1, 't%d/resbn%d' % (t, i), tape)
                acts.extend([None] * (len(tape) + 1 - len(acts)))
                acts[len(tape)] = p1
                p0_id = len(tape)
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1, c = maxpool_fwd(p1, cfg['pool_sizes'][t])
                tape.append(('maxpool', None, c))
        tapes.append((tape, p1.shape))
        flats.append(p1.reshape(p1.shape[0], -1))
    flat = np.concatenate(flats, axis=1)
    z1 = flat @ w['dense1/kernel'] + w['dense1/bias']
    h = orc.sigmoid(z1)
    z2 = h @ w['dense2/kernel'] + w['dense2/bias']
    B = flat.shape[0]
    K = cfg['output_classes']
    if K > 1:
        e = np.exp(z2 - z2.max(axis=1, keepdims=True))
        pred = e / e.sum(axis=1, keepdims=True)
        yi = np.asarray(y).astype(np.int64).reshape(-1)
        loss = float(-np.log(pred[np.arange(B), yi]).mean())
        dz2 = pred.copy()
        dz2[np.arange(B), yi] -= 1.0
        dz2 /= B
    else:
        pred = orc.sigmoid(z2)
        yt = np.asarray(y, dtype=dtype).reshape(B, 1)
        loss = float(((pred - yt) ** 2).mean())
        dz2 = 2.0 * (pred - yt) / B * pred * (1.0 - pred)
    g = {}
    g['dense2/kernel'] = h.T @ dz2
    g['dense2/bias'] = dz2.sum(axis=0)
    dh = dz2 @ w['dense2/kernel'].T
    dz1 = dh * h * (1.0 - h)
    g['dense1/kernel'] = flat.T @ dz1
    g['dense1/bias'] = dz1.sum(axis=0)
    dflat = dz1 @ w['dense1/kernel'].T
    off = 0
    for t, (tape, shp) in enumerate(tapes):
        n = int(np.prod(shp[1:]))
        d = dflat[:, off:off + n].reshape(shp)
        off += n
        pending = {}                               # tape position -> gradient arriving through a shortcut

        def run_back(tp, d):
            for pos in range(len(tp) - 1, -1, -1):
                kind, name, c = tp[pos]
                if tp is tape and (pos + 1) in pending:
                    d = d + pending.pop(pos + 1)
                if kind == 'conv':
                    d, dk, db = conv_bwd(d, c)
                    g[name + '/kernel'], g[name + '/bias'] = dk, db
                elif kind == 'bn':
                    d, dg, dbt = bn_bwd(d, c)
                    g[name + '/gamma'], g[name + '/beta'] = dg, dbt
                elif kind == 'sigmoid':
                    d = d * c * (1.0 - c)
                elif kind == 'maxpool':
                    d = maxpool_bwd(d, c)
                elif kind == 'avgpool':
                    d = avgpool_bwd(d, c)
                elif kind == 'add':
                    ds = run_back(c, d) if c else d            # shortcut branch (identity when the tape is empty)
                    pending[name] = pending.get(name, 0) + ds   # name = tape position of the shortcut source
            return d
        d_in = run_back(tape, d)
        if 0 in pending:
            d_in = d_in + pending.pop(0)
    return loss, pred, g, stats


def adagrad_step(w, g, acc, lr=0.01, eps=1e-7, initial_accumulator=0.1):
    """Adagrad on every tensor that has a gradient; returns (new weights, new accumulators).
    initial_accumulator: 0 (Keras 2.2 / tf.keras 1.13) or 0.1 (tf.keras >= 1.14)."""
    w2, acc2 = dict(w), dict(acc or {})
    for k, gk in g.items():
        a = acc2.get(k, np.full_like(gk, initial_accumulator)) + gk * gk
        acc2[k] = a
        w2[k] = (np.asarray(w[k], dtype=gk.dtype) - lr * gk / (np.sqrt(a) + eps)).astype(np.asarray(w[k]).dtype)
    return w2, acc2


def train_on_batch(w, cfg, xs, y, acc=None, lr=0.01, eps=1e-7, dtype=np.float32, initial_accumulator=0.1):
    """One training step: (loss, prediction, new weights incl. updated BN moving statistics, accumulators)."""
    loss, pred, g, stats = forward_backward(w, cfg, xs, y, dtype)
    w2, acc2 = adagrad_step(w, g, acc, lr, eps, initial_accumulator)
    for prefix, (mu, var) in stats.items():
        w2[prefix + '/mean'] = (BN_MOMENTUM * np.asarray(w[prefix + '/mean'], dtype) + (1 - BN_MOMENTUM) * mu).astype(np.float32)
        M = forward_backward.last_counts[prefix]
        bessel = dtype(M / (M - 1.0)) if M > 1 else dtype(1.0)
        w2[prefix + '/var'] = (BN_MOMENTUM * np.asarray(w[prefix + '/var'], dtype) + (1 - BN_MOMENTUM) * (var * bessel)).astype(np.float32)
    return loss, pred, w2, acc2
