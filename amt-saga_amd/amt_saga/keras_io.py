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
    for v in pools.values():
        v.sort()
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
