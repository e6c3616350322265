"""Write tests/golden/keras_unet_nfb4.hdf5 with REAL h5py / libhdf5 (this image's conda interpreter:
/opt/conda/bin/python3.9 tests/golden/make_keras_fixture.py), following keras.models.save_model of Keras 2.0.6 call by
call (attrs as utf-8 bytes, one group per layer incl. weight-less ones, datasets named by the TF variable name, Adam's
`weights` = [iterations] + ms + vs), plus the same arrays as tests/golden/keras_unet_nfb4.npz.

Keras / TensorFlow themselves are not installable here: the LAYOUT is restated from Keras' source (third party), what this
fixture pins is the CONTAINER -- deep_calcium_amd/hdf5_min.py must read what libhdf5 wrote.  Values are multiples of 1/8 so
that the files compress well in git.
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
NFB, DRP, WINDOW = 4, 0.25, 96


def graph():
    """(layer name, class, config, [(weight name, shape)]) in creation order, unet_2d_summary.py:169-222."""
    cnt, seq = {}, []

    def nm(b):
        cnt[b] = cnt.get(b, 0) + 1
        return '%s_%d' % (b, cnt[b])

    def conv(cin, cout, k=3, act='linear'):
        n = nm('conv2d')
        seq.append((n, 'Conv2D', {'filters': cout, 'kernel_size': [k, k], 'activation': act},
                    [(n + '/kernel:0', (k, k, cin, cout)), (n + '/bias:0', (cout,))]))

    def bn(c, mom):
        n = nm('batch_normalization')
        seq.append((n, 'BatchNormalization', {'momentum': mom, 'epsilon': 0.001, 'axis': -1},
                    [(n + '/' + w + ':0', (c,)) for w in ('gamma', 'beta', 'moving_mean', 'moving_variance')]))
        seq.append((nm('activation'), 'Activation', {'activation': 'relu'}, []))

    seq.append((nm('input'), 'InputLayer', {'batch_input_shape': [None, WINDOW, WINDOW], 'dtype': 'float32', 'sparse': False}, []))
    seq.append((nm('lambda'), 'Lambda', {'function': ['4wEAAAA...', None, None], 'function_type': 'lambda', 'output_shape': None,
                                          'output_shape_type': 'raw', 'arguments': {}}, []))
    cin = 1
    for lvl in range(5):
        c = NFB << lvl
        if lvl:
            seq.append((nm('max_pooling2d'), 'MaxPooling2D', {'pool_size': [2, 2], 'strides': [2, 2]}, []))
        for ci in (cin, c):
            conv(ci, c)
            bn(c, 0.99)
        if 0 < lvl < 4:
            seq.append((nm('dropout'), 'Dropout', {'rate': DRP if lvl == 1 else 2 * DRP}, []))
        cin = c
    for lvl in (3, 2, 1, 0):
        c = NFB << lvl
        n = nm('conv2d_transpose')
        seq.append((n, 'Conv2DTranspose', {'filters': c, 'kernel_size': [2, 2], 'strides': [2, 2]},
                    [(n + '/kernel:0', (2, 2, c, 2 * c)), (n + '/bias:0', (c,))]))
        bn(c, 0.5)
        seq.append((nm('dropout'), 'Dropout', {'rate': DRP if lvl == 0 else 2 * DRP}, []))
        seq.append((nm('concatenate'), 'Concatenate', {'axis': -1}, []))
        for ci in (2 * c, c):
            conv(ci, c)
            bn(c, 0.99)
    conv(NFB, 2, k=1, act='softmax')
    seq.append((nm('lambda'), 'Lambda', {'function': ['4wEAAAB...', None, None], 'function_type': 'lambda'}, []))
    return seq


def main():
    rs = np.random.RandomState(2017)
    q = lambda shape, s=1.0: (np.round(rs.standard_normal(shape) * 8 * s) / 8).astype(np.float32)
    seq = graph()
    arrays = {}
    path = os.path.join(OUT, 'keras_unet_nfb4.hdf5')
    f = h5py.File(path, 'w')
    # keras.models.save_model (2.0.6): attrs written as utf-8 encoded bytes
    f.attrs['keras_version'] = '2.0.6'.encode('utf8')
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {
        'name': 'model_1', 'layers': [{'name': n, 'class_name': c, 'config': dict(cfg, name=n), 'inbound_nodes': []}
                                      for n, c, cfg, _ in seq],
        'input_layers': [['input_1', 0, 0]], 'output_layers': [['lambda_2', 0, 0]]}}).encode('utf8')
    g = f.create_group('model_weights')
    # keras.engine.topology.save_weights_to_hdf5_group
    g.attrs['layer_names'] = [n.encode('utf8') for n, *_ in seq]
    g.attrs['backend'] = 'tensorflow'.encode('utf8')
    g.attrs['keras_version'] = '2.0.6'.encode('utf8')
    trainable = []
    k = 0
    for n, cls, cfg, ws in seq:
        lg = g.create_group(n)
        lg.attrs['weight_names'] = [w.encode('utf8') for w, _ in ws]
        for w, shp in ws:
            val = q(shp)
            if w.endswith('moving_variance:0') or w.endswith('gamma:0'):
                val = np.abs(val) + np.float32(0.5)
            d = lg.create_dataset(w, val.shape, dtype=val.dtype)
            d[...] = val
            arrays['w_%03d' % k] = val
            k += 1
            if not (w.endswith('moving_mean:0') or w.endswith('moving_variance:0')):
                trainable.append(val.shape)
    f.attrs['training_config'] = json.dumps({
        'optimizer_config': {'class_name': 'Adam', 'config': {'lr': 0.0010000000474974513, 'beta_1': 0.8999999761581421,
                                                               'beta_2': 0.9990000128746033, 'epsilon': 1e-08, 'decay': 0.0}},
        'loss': 'binary_crossentropy', 'metrics': ['F1', 'prec', 'reca', 'dice', 'dicesq', 'posyt', 'posyp'],
        'sample_weight_mode': None, 'loss_weights': None}).encode('utf8')
    og = f.create_group('optimizer_weights')
    names = ['Adam/iterations:0'] + ['training/Adam/Variable%s:0' % ('' if i == 0 else '_%d' % i) for i in range(2 * len(trainable))]
    vals = [np.array(2000.0, np.float32)] + [q(s, 0.25) for s in trainable] + [np.abs(q(s, 0.25)) for s in trainable]
    og.attrs['weight_names'] = [n.encode('utf8') for n in names]
    for n, v in zip(names, vals):
        d = og.create_dataset(n, v.shape, dtype=v.dtype)
        d[()] = v
    f.flush()
    f.close()
    arrays['opt_iterations'] = vals[0]
    for i, s in enumerate(trainable):
        arrays['opt_m_%03d' % i] = vals[1 + i]
        arrays['opt_v_%03d' % i] = vals[1 + len(trainable) + i]
    arrays['layer_names'] = np.array([n for n, *_ in seq])
    np.savez_compressed(os.path.join(OUT, 'keras_unet_nfb4.npz'), **arrays)
    print('wrote', path, os.path.getsize(path), 'bytes;', k, 'weight arrays,', len(trainable), 'trainable')


if __name__ == '__main__':
    main()
