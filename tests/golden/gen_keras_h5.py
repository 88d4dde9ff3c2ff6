#!/usr/bin/env python3
"""Write tests/golden/keras_weights_*.h5: Keras ``save_weights``-layout files produced by the real HDF5
library (libhdf5 1.10.6 found under /opt/conda in this image; TensorFlow / h5py are not available, so the
Keras side -- automatic layer names, model.layers order, attribute layout -- is restated here from
keras/engine/saving.py and the builder of RDCNN.py:176-233, independently of the product's importer).

    python tests/golden/gen_keras_h5.py        (this container only; needs gcc + /opt/conda/lib/libhdf5)

Two small topologies (every layer kind: Cin = 1 conv, projected and identity shortcuts, pooling, two towers)
with weights drawn by res_net.init_weights(seed, calibrated=False); the expected canonical dict is the draw
itself, so the test is: file -> importer == draw, bit for bit.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
from amt_saga.rdcnn import res_net          # noqa: E402

CASES = {
    'shallow': dict(input_shapes=[(12, 10, 1)], output_classes=1, output_range=[3, 40], kernel_sizes=[(4, 2)],
                    pool_sizes=[(2, 2)], convolutional_layer_count=5, feature_expand_frequency=2,
                    pool_layer_frequency=2, residual_layer_frequencies=2, weight_seed=77),
    'dual': dict(input_shapes=[(16, 8, 1), (16, 8, 1)], output_classes=7, output_range=[0, 1],
                 kernel_sizes=[(2, 2), (2, 2)], pool_sizes=[(2, 2), (2, 2)], convolutional_layer_count=4,
                 feature_expand_frequency=2, pool_layer_frequency=2, residual_layer_frequencies=2, weight_seed=78),
}


def keras_graph(cfg):
    """Simulate the Keras functional graph of RDCNN.py:176-233: [(layer name, class, canonical prefix or None,
    depth-from-output)] -- names by per-class creation counters, depth = longest path to the output."""
    counters = {}
    nodes = []          # (name, cls, prefix, inputs[list of node idx])

    def new(cls, prefix, inputs):
        counters[cls] = counters.get(cls, 0) + 1
        nodes.append(('%s_%d' % (cls, counters[cls]), cls, prefix, list(inputs)))
        return len(nodes) - 1

    r = cfg['residual_layer_frequencies'][0] if cfg['residual_layer_frequencies'] else 0
    tails = []
    for t, (H, W, _) in enumerate(cfg['input_shapes']):
        p1 = new('input', None, [])
        ph, pw = cfg['pool_sizes'][t]
        C, fo = 1, 32
        p0, p0s = p1, (H, W, 1)
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            p1 = new('conv2d', 't%d/conv%d' % (t, i), [p1])
            p1 = new('batch_normalization', 't%d/bn%d' % (t, i), [p1])
            p1 = new('activation', None, [p1])
            C = fo
            if r and i % r == 0:
                a = p0
                if p0s != (H, W, C):
                    if p0s[2] != C:
                        a = new('conv2d', 't%d/sc%d' % (t, i), [a])
                    if p0s[:2] != (H, W):
                        a = new('average_pooling2d', None, [a])
                    a = new('batch_normalization', 't%d/scbn%d' % (t, i), [a])
                p1 = new('add', None, [a, p1])
                p1 = new('batch_normalization', 't%d/resbn%d' % (t, i), [p1])
                p0, p0s = p1, (H, W, C)
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1 = new('max_pooling2d', None, [p1])
                H, W = H // ph, W // pw
            if cfg['feature_expand_frequency'] and i % cfg['feature_expand_frequency'] == 0:
                fo *= 2
        tails.append(new('flatten', None, [p1]))
    m = tails[0] if len(tails) == 1 else new('concatenate', None, tails)
    m = new('dense', 'dense1', [m])
    m = new('activation', None, [m])
    m = new('dense', 'dense2', [m])
    out = new('activation', None, [m])
    depth = [0] * len(nodes)
    for idx in range(len(nodes) - 1, -1, -1):            # creation order is a topological order
        for j in nodes[idx][3]:
            depth[j] = max(depth[j], depth[idx] + 1)
    # model.layers: by depth; layers of equal depth by Keras' traversal index -- _map_graph_network's build_map numbers
    # a layer the first time a depth-first walk from the output reaches it, descending into a node's inbound layers
    # in the order they were passed to the call (Add()([intermediate, layer_to]): the shortcut branch first).
    # Written recursively here, on purpose differently from the product's exporter (amt_saga/keras_io.keras_layers).
    seen = {}

    def build_map(k):
        if k in seen:
            return
        seen[k] = len(seen)
        for j in nodes[k][3]:
            build_map(j)
    sys.setrecursionlimit(10000)
    build_map(out)
    order = sorted(range(len(nodes)), key=lambda k: (-depth[k], seen[k]))
    return [nodes[k] for k in order]


def main():
    src = os.path.join(HERE, 'keras_h5_writer.c')
    exe = os.path.join(tempfile.gettempdir(), 'keras_h5_writer')
    subprocess.check_call(['gcc', '-I/opt/conda/include', src, '-L/opt/conda/lib', '-lhdf5',
                           '-Wl,-rpath,/opt/conda/lib', '-o', exe])
    for tag, kw in CASES.items():
        net = res_net(calibrated=False, **kw)
        w = net.weights
        blob, lines, off = [], [], 0
        for name, cls, prefix, _ in keras_graph(net.cfg):
            lines.append('layer %s' % name)
            if prefix is None:
                continue
            keys = (('kernel', 'kernel'), ('bias', 'bias')) if cls in ('conv2d', 'dense') else \
                (('gamma', 'gamma'), ('beta', 'beta'), ('moving_mean', 'mean'), ('moving_variance', 'var'))
            for kname, ours in keys:
                a = np.ascontiguousarray(w[prefix + '/' + ours], dtype=np.float32)
                lines.append('weight %s/%s:0 %d %d %s' % (name, kname, off, a.ndim, ' '.join(map(str, a.shape))))
                blob.append(a.reshape(-1))
                off += a.size
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, 'm.txt'), 'w').write('\n'.join(lines) + '\n')
            np.concatenate(blob).tofile(os.path.join(d, 'w.bin'))
            out = os.path.join(HERE, 'keras_weights_%s.h5' % tag)
            subprocess.check_call([exe, os.path.join(d, 'm.txt'), os.path.join(d, 'w.bin'), out])
        print(tag, os.path.getsize(out), 'bytes;', sum(1 for l in lines if l.startswith('layer')), 'layers,',
              sum(1 for l in lines if l.startswith('weight')), 'weight tensors')


if __name__ == '__main__':
    main()
