"""Keras-2.0.x HDF5 model files, read and written in-process (deep_calcium_amd/hdf5_min.py + keras_io.py; SURVEY 8f
rank 1).  The fixture tests/golden/keras_unet_nfb4.hdf5 was written by REAL h5py / libhdf5 following Keras 2.0.6's
save_model call by call (tests/golden/make_keras_fixture.py): it pins the container.  The layout itself is restated from
Keras' source (Keras is not installable offline: third-party parity unpinned, stated in DESIGN.md)."""
import json
import os
import subprocess

import numpy as np
import pytest

CONDA_PY = '/opt/conda/bin/python3.9'


def _have_h5py():
    return os.path.exists(CONDA_PY) and subprocess.run([CONDA_PY, '-c', 'import h5py'], capture_output=True).returncode == 0


def test_reads_libhdf5_written_keras_file(golden_dir):
    from deep_calcium_amd import keras_io
    z = np.load(os.path.join(golden_dir, 'keras_unet_nfb4.npz'))
    st = keras_io.read_keras_model(os.path.join(golden_dir, 'keras_unet_nfb4.hdf5'))
    assert len(st['weights']) == 134
    assert all(np.array_equal(w, z['w_%03d' % i]) and w.dtype == np.float32 for i, w in enumerate(st['weights']))
    assert st['config'] == dict(window_shape=(96, 96), nb_filters_base=4, prop_dropout_base=0.25, upsampling_or_transpose='transpose')
    assert st['loss'] == 'binary_crossentropy'
    opt = st['optimizer']
    assert opt['iterations'] == 2000 and len(opt['m']) == 90 and len(opt['v']) == 90
    assert abs(opt['config']['lr'] - 0.001) < 1e-9 and abs(opt['config']['beta_2'] - 0.999) < 1e-6
    for i in range(90):
        assert np.array_equal(opt['m'][i], z['opt_m_%03d' % i]) and np.array_equal(opt['v'][i], z['opt_v_%03d' % i])


def test_container_details(golden_dir):
    """hdf5_min on the same file: attributes (fixed-length byte strings, string arrays), nested groups, scalars."""
    from deep_calcium_amd import hdf5_min
    f = hdf5_min.File(os.path.join(golden_dir, 'keras_unet_nfb4.hdf5'))
    # h5py stores Python bytes as VARIABLE-length ASCII strings (global heap): they come back as text
    assert f.attrs['keras_version'] == '2.0.6' and f.attrs['backend'] == 'tensorflow'
    cfg = json.loads(f.attrs['model_config'])
    assert cfg['config']['layers'][0]['config']['batch_input_shape'] == [None, 96, 96]
    names = [n if isinstance(n, str) else n.decode() for n in f['model_weights'].attrs['layer_names']]
    assert names[:4] == ['input_1', 'lambda_1', 'conv2d_1', 'batch_normalization_1'] and names[-1] == 'lambda_2'
    assert len(f['model_weights/activation_1'].attrs['weight_names']) == 0
    d = f['model_weights/conv2d_transpose_2/conv2d_transpose_2/kernel:0']
    assert d.shape == (2, 2, 16, 32) and d.dtype == np.float32
    assert f['optimizer_weights/Adam/iterations:0'].shape == () and float(f['optimizer_weights/Adam/iterations:0'].read()) == 2000.0
    assert sorted(f.keys()) == ['model_weights', 'optimizer_weights']
    with pytest.raises(KeyError):
        f['model_weights/nope']
    assert hdf5_min.is_hdf5(os.path.join(golden_dir, 'keras_unet_nfb4.hdf5')) and not hdf5_min.is_hdf5(os.path.join(golden_dir, 'tta.npz'))


@pytest.mark.parametrize('upsampling', [False, True])
def test_write_then_read_round_trip(tmp_path, upsampling):
    from deep_calcium_amd import keras_io
    from oracle import unet_numpy as on
    nfb = 8
    W = on.init_weights(nfb, seed=3, randomize_bn=True, upsampling=upsampling)
    cfg = dict(window_shape=(128, 128), nb_filters_base=nfb, prop_dropout_base=0.125,
               upsampling_or_transpose='upsampling' if upsampling else 'transpose')
    rs = np.random.RandomState(0)
    shapes = [s for i, s in enumerate(on.weight_shapes(nfb, upsampling))]
    # trainable = everything but the BN moving statistics (positions 4, 5 of each 6-array layer block)
    tr, i = [], 0
    for name, kind, cin, cout, mom in on.layer_table(nfb, upsampling):
        n = 2 if kind == 'head' else 6
        tr += shapes[i:i + min(n, 4)]
        i += n
    opt = dict(config=dict(lr=0.0005, beta_1=0.9, beta_2=0.999, epsilon=1e-8), iterations=321,
               m=[rs.standard_normal(s).astype(np.float32) for s in tr], v=[rs.random_sample(s).astype(np.float32) for s in tr])
    path = str(tmp_path / 'model_03_0.512.hdf5')
    keras_io.write_keras_model(path, W, cfg, optimizer=opt, loss='dice_loss', metrics=['F1'])
    st = keras_io.read_keras_model(path)
    assert st['config'] == cfg and st['loss'] == 'dice_loss'
    assert len(st['weights']) == len(W) and all(np.array_equal(a, np.asarray(b, np.float32)) for a, b in zip(st['weights'], W))
    assert st['optimizer']['iterations'] == 321 and abs(st['optimizer']['config']['lr'] - 0.0005) < 1e-12
    assert all(np.array_equal(a, b) for a, b in zip(st['optimizer']['m'], opt['m']))
    assert all(np.array_equal(a, b) for a, b in zip(st['optimizer']['v'], opt['v']))
    # weights only (model.save(include_optimizer=False) / an uncompiled model)
    keras_io.write_keras_model(path, W, cfg)
    st = keras_io.read_keras_model(path)
    assert st['optimizer'] is None and len(st['weights']) == len(W)
    with pytest.raises(ValueError):
        keras_io.write_keras_model(path, W[:-1], cfg)


def test_written_file_is_read_by_real_h5py_the_way_keras_does(tmp_path):
    """keras.engine.topology.load_weights_from_hdf5_group's walk (layer_names -> weight_names -> datasets), executed by
    real h5py on a file written by hdf5_min: the arrays come back in get_weights() order."""
    if not _have_h5py():
        pytest.skip('no interpreter with h5py')
    from deep_calcium_amd import keras_io
    from oracle import unet_numpy as on
    nfb = 4
    W = on.init_weights(nfb, seed=9, randomize_bn=True)
    path, ref, out = str(tmp_path / 'm.hdf5'), str(tmp_path / 'ref.npz'), str(tmp_path / 'got.npz')
    keras_io.write_keras_model(path, W, dict(window_shape=(64, 64), nb_filters_base=nfb, prop_dropout_base=0.25,
                                            upsampling_or_transpose='transpose'))
    reader = '''
import h5py, json, numpy as np, sys
f = h5py.File(sys.argv[1], 'r')
g = f['model_weights']
layer_names = [n.decode('utf8') for n in g.attrs['layer_names']]
filtered = [n for n in layer_names if len(g[n].attrs['weight_names'])]
out = []
for n in filtered:
    for w in g[n].attrs['weight_names']:
        out.append(np.asarray(g[n][w.decode('utf8')]))
cfg = json.loads(f.attrs['model_config'].decode('utf8'))
assert f.attrs['keras_version'] == b'2.0.6' and cfg['class_name'] == 'Model'
assert cfg['config']['layers'][0]['config']['batch_input_shape'] == [None, 64, 64]
seen = []
f.visit(seen.append)
np.savez(sys.argv[2], *out)
print(len(filtered), len(out), len(seen))
'''
    r = subprocess.run([CONDA_PY, '-c', reader, path, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[:2] == ['45', '134']
    z = np.load(out)
    assert all(np.array_equal(z['arr_%d' % i], np.asarray(w, np.float32)) for i, w in enumerate(W))


def test_unsupported_files_fail_loudly(tmp_path):
    from deep_calcium_amd import hdf5_min, keras_io
    from deep_calcium_amd.model import read_checkpoint
    p = str(tmp_path / 'junk.hdf5')
    open(p, 'wb').write(b'not a model')
    with pytest.raises(ValueError):
        read_checkpoint(p)
    w = hdf5_min.Writer()
    w.create_dataset('x', np.zeros(3, np.float32))
    p2 = str(tmp_path / 'other.hdf5')
    w.save(p2)
    with pytest.raises(ValueError):
        keras_io.read_keras_model(p2)
    with pytest.raises(hdf5_min.Hdf5Error):
        big = hdf5_min.Writer()
        big.attrs['model_config'] = b'x' * 70000          # v1 object-header messages stop at 64 KiB, as in h5py
        big.save(str(tmp_path / 'big.hdf5'))


def test_layout_check_names_the_first_difference():
    """keras_io.check_keras_layout: a genuine Keras file is mapped by position, so its weighted layers are verified against
    the UNet2DS graph and the FIRST difference raises with the layer's name (never a silent mis-map)."""
    from deep_calcium_amd import keras_io
    seq = keras_io.keras_layer_sequence(8)

    def lists(mut=None, renumber=0):
        ln, wn, sh = [], [], []
        for n, cls, _, ws in seq:
            if renumber:                                  # a model built second in a session: conv2d_24 ...
                head, _, tail = n.rpartition('_')
                n = '%s_%d' % (head, int(tail) + renumber)
            ln.append(n)
            wn.append(['%s/%s' % (n, sfx) for sfx, _ in ws])
            sh.append([s for _, s in ws])
        if mut:
            mut(ln, wn, sh)
        return ln, wn, sh
    assert keras_io.check_keras_layout('f', *lists()) == (8, False)
    assert keras_io.check_keras_layout('f', *lists(renumber=23)) == (8, False)
    k_bn = [i for i, (n, *_) in enumerate(seq) if n == 'batch_normalization_3'][0]

    def swap(ln, wn, sh):
        wn[k_bn][0], wn[k_bn][1] = wn[k_bn][1], wn[k_bn][0]
    with pytest.raises(ValueError, match=r"batch_normalization_3.*\['gamma', 'beta', 'moving_mean', 'moving_variance'\]"):
        keras_io.check_keras_layout('f', *lists(swap))

    def drop(ln, wn, sh):
        del ln[k_bn], wn[k_bn], sh[k_bn]
    with pytest.raises(ValueError, match=r"is 'conv2d_4' where the UNet2DS graph .* has a BatchNormalization \('batch_normalization_3'\)"):
        keras_io.check_keras_layout('f', *lists(drop))

    def reshape(ln, wn, sh):
        k = [i for i, (n, *_) in enumerate(seq) if n == 'conv2d_transpose_2'][0]
        sh[k][0] = sh[k][0][:2] + (sh[k][0][3], sh[k][0][2])          # (2,2,Cin,Cout) instead of Keras' (2,2,Cout,Cin)
    with pytest.raises(ValueError, match=r"conv2d_transpose_2/kernel:0' has shape"):
        keras_io.check_keras_layout('f', *lists(reshape))

    def truncate(ln, wn, sh):
        del ln[-2:], wn[-2:], sh[-2:]
    with pytest.raises(ValueError, match='the file ends after 44 weighted layers'):
        keras_io.check_keras_layout('f', *lists(truncate))
