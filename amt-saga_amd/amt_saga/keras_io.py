"""Import of the reference's Keras checkpoints (``model.save_weights`` HDF5, RDCNN.py:490-494; resumed by
``load_weights``, RDCNN.py:778-782, training.py:104-138) into the canonical weight dictionary of
``res_net`` (names "t0/conv1/kernel", ... -- see rdcnn.py).

Keras identifies weights by layer name; the builder of RDCNN.py:176-233 names nothing, so the layers carry
Keras' automatic names ``<class>_<n>`` with one counter per class, incremented in CREATION order.  The file
lists the layers in ``model.layers`` order (sorted by graph depth, where the shortcut branch -- 1x1 Conv2D,
AveragePooling2D, BatchNormalization, RDCNN.py:328-334 -- interleaves with the main branch), so the importer
does not rely on the file order: it sorts the layers of each class by their counter and deals them out along
the same walk of the builder that defines the canonical order: per tower and conv layer i
    Conv2D, BatchNormalization, [at a shortcut layer: Conv2D 1x1 if channels differ, BatchNormalization if
    shapes differ (the AveragePooling2D between them has no weights), BatchNormalization after the Add]
then Dense(300), Dense(K).  Every array's shape is checked against the topology
(``ValueError('Invalid Input shape. ...')`` otherwise).
"""
import re

import numpy as np

from . import hdf5

_BN_KEYS = (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'mean'), ('moving_variance', 'var'))


def creation_plan(layout):
    """[(keras class, canonical prefix)] of the weighted layers in creation order, from res_net.layout."""
    plan, seen = [], set()
    for name, _ in layout:
        prefix = name.rsplit('/', 1)[0]
        if prefix in seen:
            continue
        seen.add(prefix)
        leaf = prefix.rsplit('/', 1)[-1]
        if leaf.startswith('conv') or leaf.startswith('sc') and not leaf.startswith('scbn'):
            plan.append(('conv2d', prefix))
        elif leaf.startswith('dense'):
            plan.append(('dense', prefix))
        else:
            plan.append(('batch_normalization', prefix))
    return plan


def _counter(layer_name):
    m = re.search(r'_(\d+)$', layer_name)
    return int(m.group(1)) if m else 0          # tf.keras 2.x names the first instance without a suffix


def _layer_class(layer_name):
    return re.sub(r'_\d+$', '', layer_name)


def read_layers(path):
    """{layer name: {weight leaf name: array}} of a Keras weights file, plus the layer list in file order."""
    f = hdf5.File(path)
    root = f.attrs('/')
    names = root.get('layer_names')
    if names is None:                            # Keras splits oversized attributes into name0, name1, ...
        parts, i = [], 0
        while 'layer_names%d' % i in root:
            parts.append(root['layer_names%d' % i])
            i += 1
        if not parts:
            # a full model file (model.save) keeps the same structure under /model_weights
            if 'model_weights' in f.keys('/'):
                raise ValueError('Keras full-model file: pass the weights group (save_weights format expected)')
            raise ValueError('not a Keras weights file: no layer_names attribute')
        names = np.concatenate(parts)
    layers = [n.decode('utf8') if isinstance(n, bytes) else str(n) for n in np.atleast_1d(names)]
    out = {}
    for ln in layers:
        wn = f.attrs('/' + ln).get('weight_names')
        arrays = {}
        if wn is not None and np.size(wn):
            for w in np.atleast_1d(wn):
                w = w.decode('utf8') if isinstance(w, bytes) else str(w)
                leaf = w.rsplit('/', 1)[-1].split(':')[0]
                arrays[leaf] = f.dataset('/' + ln + '/' + w)
        out[ln] = arrays
    return out, layers


def load_keras_weights(path, layout):
    """Canonical weight dict for a res_net with `layout` from a Keras ``save_weights`` file."""
    by_layer, _ = read_layers(path)
    pools = {}
    for ln, arrays in by_layer.items():
        if arrays:
            pools.setdefault(_layer_class(ln), []).append((_counter(ln), ln))
    # Keras' name counters are per process, not per model: a model built after another one in the same process
    # starts at conv2d_34 / batch_normalization_53 / dense_3 (and tf.keras 2.x names the first instance without a
    # suffix).  Only the ORDER of the counters inside a class matters, so offsets and gaps are accepted; what cannot
    # be ordered is rejected: two weighted layers of one class with the same counter, or a weighted layer class the
    # builder of RDCNN.py:176-233 never creates.
    known = {'conv2d', 'batch_normalization', 'dense'}
    for cls, v in pools.items():
        if cls not in known:
            raise ValueError('Invalid Input shape. Expected: layers of {} . Got: weighted layer(s) {}'.format(
                sorted(known), ', '.join(ln for _, ln in v[:4])))
        v.sort()
        dup = [b for a, b in zip(v, v[1:]) if a[0] == b[0]]
        if dup:
            raise ValueError('Invalid Input shape. Expected: distinct name counters per layer class . Got: {} twice '
                             '(ambiguous creation order)'.format(dup[0][1]))
    shapes = dict(layout)
    w = {}

    def put(name, a, layer):
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shapes[name]):
            raise ValueError('Invalid Input shape. Expected: {} . Got: {} ({} <- {})'.format(
                tuple(shapes[name]), tuple(a.shape), name, layer))
        w[name] = a

    for cls, prefix in creation_plan(layout):
        if not pools.get(cls):
            raise ValueError('Invalid Input shape. Expected: a {} layer for {} . Got: none left'.format(cls, prefix))
        _, ln = pools[cls].pop(0)
        arrays = by_layer[ln]
        if cls == 'batch_normalization':
            for kk, ours in _BN_KEYS:
                put(prefix + '/' + ours, arrays[kk], ln)
        else:
            put(prefix + '/kernel', arrays['kernel'], ln)
            put(prefix + '/bias', arrays['bias'], ln)
    left = [ln for v in pools.values() for _, ln in v]
    if left:
        raise ValueError('Invalid Input shape. Expected: {} weighted layers . Got: {} more ({})'.format(
            len(creation_plan(layout)), len(left), ', '.join(left[:4])))
    return w


# ---- export: the same file Keras' save_weights would write for this graph -------------------------------
def keras_layers(cfg):
    """[(layer name, canonical prefix or None)] in ``model.layers`` order for the graph RDCNN.py:176-233 builds:
    names ``<class>_<n>`` from per-class creation counters (Keras' automatic naming), order by graph depth
    (longest path to the output), and -- as keras.engine.network._map_graph_network does -- layers of EQUAL depth by
    their traversal index: a pre-order depth-first walk from the output over each layer's inbound layers in call
    order.  ``Add()([intermediate, layer_to])`` (RDCNN.py:316,335) lists the shortcut branch first, so at a projected
    shortcut the 1x1 Conv2D precedes the main branch's Conv2D of equal depth and tower 0 precedes tower 1
    (``Concatenate()(layers)``).  Round 2 broke ties by creation order, which swaps those two convolutions.
    Keras itself is not available here: this order is restated from its source, not pinned by a Keras-written file."""
    counters, nodes = {}, []

    def new(cls, prefix, inputs):
        counters[cls] = counters.get(cls, 0) + 1
        nodes.append(('%s_%d' % (cls, counters[cls]), prefix, list(inputs)))
        return len(nodes) - 1

    rf = cfg['residual_layer_frequencies']
    r = rf[0] if rf else 0
    tails = []
    for t, (H, W, _) in enumerate(cfg['input_shapes']):
        cur = new('input', None, [])
        ph, pw = cfg['pool_sizes'][t]
        C, fo = 1, 32
        p0, p0_shape = cur, (H, W, 1)
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            cur = new('conv2d', 't%d/conv%d' % (t, i), [cur])
            cur = new('batch_normalization', 't%d/bn%d' % (t, i), [cur])
            cur = new('activation', None, [cur])
            C = fo
            if r and i % r == 0:
                a = p0
                if p0_shape != (H, W, C):
                    if p0_shape[2] != C:
                        a = new('conv2d', 't%d/sc%d' % (t, i), [a])
                    if p0_shape[:2] != (H, W):
                        a = new('average_pooling2d', None, [a])
                    a = new('batch_normalization', 't%d/scbn%d' % (t, i), [a])
                cur = new('add', None, [a, cur])
                cur = new('batch_normalization', 't%d/resbn%d' % (t, i), [cur])
                p0, p0_shape = cur, (H, W, C)
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                cur = new('max_pooling2d', None, [cur])
                H, W = H // ph, W // pw
            if cfg['feature_expand_frequency'] and i % cfg['feature_expand_frequency'] == 0:
                fo *= 2
        tails.append(new('flatten', None, [cur]))
    m = tails[0] if len(tails) == 1 else new('concatenate', None, tails)
    m = new('dense', 'dense1', [m])
    m = new('activation', None, [m])
    m = new('dense', 'dense2', [m])
    new('activation', None, [m])
    depth = [0] * len(nodes)
    for idx in range(len(nodes) - 1, -1, -1):
        for j in nodes[idx][2]:
            depth[j] = max(depth[j], depth[idx] + 1)
    # traversal index: pre-order DFS from the output, inbound layers in call order (explicit stack: no recursion limit)
    index, stack = {}, [len(nodes) - 1]
    while stack:
        k = stack.pop()
        if k in index:
            continue
        index[k] = len(index)
        stack.extend(reversed(nodes[k][2]))
    order = sorted(range(len(nodes)), key=lambda k: (-depth[k], index[k]))
    return [(nodes[k][0], nodes[k][1]) for k in order]


def save_keras_weights(path, weights, cfg, keras_version=b'2.2.4-tf', backend=b'tensorflow'):
    """Write `weights` (canonical dict) as the HDF5 file ``model.save_weights(path)`` produces for the reference's
    graph (RDCNN.py:490-494): root attributes layer_names / backend / keras_version, one group per layer with its
    weight_names, datasets <layer>/<layer>/<weight>:0 -- so that Keras' ``load_weights`` (RDCNN.py:778-782) and
    this build's importer both take it."""
    w = hdf5.Writer()
    layers = keras_layers(cfg)
    w.group('/', {'layer_names': [n.encode('utf8') for n, _ in layers], 'backend': backend, 'keras_version': keras_version})
    for name, prefix in layers:
        if prefix is None:
            w.group('/' + name, {'weight_names': []})
            continue
        if name.startswith('batch_normalization'):
            keys = (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'mean'), ('moving_variance', 'var'))
        else:
            keys = (('kernel', 'kernel'), ('bias', 'bias'))
        w.group('/' + name, {'weight_names': [('%s/%s:0' % (name, k)).encode('utf8') for k, _ in keys]})
        for k, ours in keys:
            w.dataset('/%s/%s/%s:0' % (name, name, k), np.asarray(weights[prefix + '/' + ours], dtype=np.float32))
    w.save(path)

