"""Keras-2.0.x model files (HDF5) <-> the HIP UNet2DS model: the reference's checkpoint format, read and written in-process.

What the reference does with these files: `ModelCheckpoint` saves them every epoch
(/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:423-424), `fit(model_path=...)` / `predict(model_path=...)`
load them through `load_model_with_new_input_shape` (/root/reference/deepcalcium/utils/keras_helpers.py:24-68), and the
released `unet2ds_model.hdf5` (:28) is one.  File layout = keras.models.save_model of Keras 2.0.6 (un-vendored third party,
restated; the container itself is read / written by hdf5_min.py, no h5py needed):

    /  attrs: keras_version, backend, model_config (JSON), training_config (JSON, compiled models)
    /model_weights            attrs: layer_names [bytes], backend, keras_version
    /model_weights/<layer>    attrs: weight_names [bytes]; one dataset per weight, named '<layer>/kernel:0' etc.
                              (so the dataset sits in a nested group: model_weights/conv2d_1/conv2d_1/kernel:0)
    /optimizer_weights        attrs: weight_names; datasets 'Adam/iterations:0' + one per Adam m, then one per Adam v
                              (Keras 2.0.6: Adam.weights = [iterations] + ms + vs, in model.trainable_weights order)

Concatenating the per-layer datasets in (layer_names, weight_names) order IS `model.get_weights()`: the 134 arrays (110
with UpSampling2D) of UNetEngine.get_weights().  trainable_weights order = kernel, bias, gamma, beta per layer = the flat
parameter buffer's order, so Adam's m / v map onto mflat / vflat by concatenation.

Limits, stated: a file written here carries a `model_config` that describes the graph (layer classes, names, shapes) but
not the two Lambda layers' marshalled Python bytecode, so Keras' `load_model` cannot rebuild the graph from it --
`unet(...)` + `model.load_weights(path)` (by topology) is the Keras-side entry; `load_model_with_new_input_shape` of THIS
package reads both its own files and genuine Keras files.
"""
import json

import numpy as np

from . import hdf5_min

KERAS_VERSION = b'2.0.6'


def _s(v):
    return v.decode('utf8') if isinstance(v, (bytes, np.bytes_)) else str(v)


# ---- the reference graph as Keras would name it (unet_2d_summary.py:169-222, fresh session: indices from 1) ---------------
def keras_layer_sequence(nfb=32, drp=0.25, upsampling=False, window=(128, 128)):
    """[(keras layer name, class name, config dict, [(weight suffix, shape)])] in graph-creation order."""
    cnt = {}

    def nm(base):
        cnt[base] = cnt.get(base, 0) + 1
        return '%s_%d' % (base, cnt[base])

    seq = []

    def conv_layer(cin, cout):
        n = nm('conv2d')
        seq.append((n, 'Conv2D', dict(filters=cout, kernel_size=[3, 3], strides=[1, 1], padding='same', activation='linear'),
                    [('kernel:0', (3, 3, cin, cout)), ('bias:0', (cout,))]))
        bn(cout, 0.99)

    def bn(c, momentum):
        n = nm('batch_normalization')
        seq.append((n, 'BatchNormalization', dict(axis=-1, momentum=momentum, epsilon=0.001),
                    [('gamma:0', (c,)), ('beta:0', (c,)), ('moving_mean:0', (c,)), ('moving_variance:0', (c,))]))
        seq.append((nm('activation'), 'Activation', dict(activation='relu'), []))

    def up_layer(cin, cout):
        if upsampling:
            seq.append((nm('up_sampling2d'), 'UpSampling2D', dict(size=[2, 2]), []))
            return
        n = nm('conv2d_transpose')
        seq.append((n, 'Conv2DTranspose', dict(filters=cout, kernel_size=[2, 2], strides=[2, 2], padding='valid'),
                    [('kernel:0', (2, 2, cout, cin)), ('bias:0', (cout,))]))
        bn(cout, 0.5)

    seq.append((nm('input'), 'InputLayer', dict(batch_input_shape=[None, int(window[0]), int(window[1])], dtype='float32'), []))
    seq.append((nm('lambda'), 'Lambda', dict(output_shape=None, note='K.expand_dims(x, axis=-1)'), []))
    cin = 1
    for lvl in range(5):
        c = nfb << lvl
        if lvl:
            seq.append((nm('max_pooling2d'), 'MaxPooling2D', dict(pool_size=[2, 2], strides=[2, 2], padding='valid'), []))
        conv_layer(cin, c)
        conv_layer(c, c)
        if 0 < lvl < 4:
            seq.append((nm('dropout'), 'Dropout', dict(rate=drp if lvl == 1 else 2 * drp), []))
        cin = c
    for lvl in (3, 2, 1, 0):
        c = nfb << lvl
        up_layer(2 * c, c)
        seq.append((nm('dropout'), 'Dropout', dict(rate=drp if lvl == 0 else 2 * drp), []))
        seq.append((nm('concatenate'), 'Concatenate', dict(axis=-1), []))
        conv_layer(3 * c if upsampling else 2 * c, c)
        conv_layer(c, c)
    n = nm('conv2d')
    seq.append((n, 'Conv2D', dict(filters=2, kernel_size=[1, 1], strides=[1, 1], padding='valid', activation='softmax'),
                [('kernel:0', (1, 1, nfb, 2)), ('bias:0', (2,))]))
    seq.append((nm('lambda'), 'Lambda', dict(output_shape=None, note='x[:, :, :, -1]'), []))
    return seq


# ---- reading ------------------------------------------------------------------------------------------------------------
def _base(name):
    """'conv2d_transpose_17' -> 'conv2d_transpose' (Keras numbers layers per session: a model built second is conv2d_24...)."""
    head, _, tail = name.rpartition('_')
    return head if head and tail.isdigit() else name


def check_keras_layout(path, layer_names, weight_names, shapes):
    """First contact with a file this build did not write (the released unet2ds_model.hdf5, unet_2d_summary.py:28) must not
    be a silent mis-map: the arrays are taken in (layer_names, weight_names) order and would land on whatever layer the flat
    buffer has at that position.  So the file's WEIGHTED layers are checked against the UNet2DS graph (unet_2d_summary.py:
    169-222) one by one -- layer kind (the name's base: conv2d / batch_normalization / conv2d_transpose; the per-session
    numbering is free), the weights' names (kernel, bias / gamma, beta, moving_mean, moving_variance, in that order) and
    their shapes -- and the first difference raises, naming the layer and what was expected there.  Weightless layers
    (Activation, Dropout, MaxPooling2D, Lambda ...) carry nothing to map and are not compared."""
    got = [(ln, wn, sh) for ln, wn, sh in zip(layer_names, weight_names, shapes) if len(wn)]
    if not got:
        raise ValueError('%s: no layer with weights' % path)
    first = got[0][2][0] if got[0][2] else ()
    if len(first) != 4 or tuple(first[:3]) != (3, 3, 1):
        raise ValueError('%s: first weighted layer %r should hold the (3,3,1,nfb) kernel of the first Conv2D, found shapes %r'
                         % (path, got[0][0], got[0][2]))
    nfb = int(first[-1])
    n_arrays = sum(len(wn) for _, wn, _ in got)
    ups = n_arrays == 110
    want = [(n, cls, ws) for n, cls, _, ws in keras_layer_sequence(nfb, 0.25, ups) if ws]
    for k in range(max(len(got), len(want))):
        if k >= len(got):
            raise ValueError('%s: the file ends after %d weighted layers; the UNet2DS graph continues with %s %r'
                             % (path, len(got), want[k][1], want[k][0]))
        ln, wn, sh = got[k]
        if k >= len(want):
            raise ValueError('%s: unexpected extra weighted layer %r (%s) after the %d of the UNet2DS graph'
                             % (path, ln, ', '.join(wn), len(want)))
        en, ecls, ews = want[k]
        if _base(ln) != _base(en):
            raise ValueError('%s: weighted layer %d is %r where the UNet2DS graph (nb_filters_base %d, %s) has a %s (%r)'
                             % (path, k, ln, nfb, 'UpSampling2D' if ups else 'Conv2DTranspose', ecls, en))
        leaf = [w.rsplit('/', 1)[-1].split(':')[0] for w in wn]
        eleaf = [sfx.split(':')[0] for sfx, _ in ews]
        if leaf != eleaf:
            raise ValueError('%s: layer %r lists weights %r where %s has %r (in this order)' % (path, ln, list(wn), ecls, eleaf))
        for w, a, (sfx, es) in zip(wn, sh, ews):
            if tuple(a) != tuple(es):
                raise ValueError('%s: %r has shape %r, the UNet2DS graph has %r there (%s of %r)' % (path, w, tuple(a), tuple(es), sfx, en))
    return nfb, ups


def read_keras_model(path):
    """-> dict(weights=[...get_weights() order], config=dict(window_shape, nb_filters_base, prop_dropout_base,
    upsampling_or_transpose), optimizer=None | dict(config=..., iterations, m=[...], v=[...]), loss=str | None)."""
    f = hdf5_min.File(path)
    g = f['model_weights'] if 'model_weights' in f else f          # save_weights() files have no wrapper group
    if 'layer_names' not in g.attrs:
        raise ValueError('%s: no layer_names attribute -- not a Keras model / weights file' % path)
    weights = []
    lnames, wnames, shapes = [], [], []
    for lname in np.atleast_1d(g.attrs['layer_names']):
        lg = g[_s(lname)]
        names = [_s(w) for w in np.atleast_1d(lg.attrs.get('weight_names', []))]
        arrs = [np.asarray(lg[w].read(), dtype=np.float32) for w in names]
        weights += arrs
        lnames.append(_s(lname))
        wnames.append(names)
        shapes.append([a.shape for a in arrs])
    check_keras_layout(path, lnames, wnames, shapes)
    if len(weights) not in (134, 110):
        raise ValueError('%s: expected 134 (Conv2DTranspose) or 110 (UpSampling2D) weight arrays of a UNet2DS model, found %d'
                         % (path, len(weights)))
    if weights[0].ndim != 4 or weights[0].shape[:3] != (3, 3, 1):
        raise ValueError('%s: first array should be the (3,3,1,nfb) kernel, got %r' % (path, weights[0].shape))
    config = dict(window_shape=(512, 512), nb_filters_base=int(weights[0].shape[-1]), prop_dropout_base=0.25,
                  upsampling_or_transpose='transpose' if len(weights) == 134 else 'upsampling')
    mc = f.attrs.get('model_config')
    if mc is not None:
        try:
            layers = json.loads(_s(mc))['config']['layers']
        except (ValueError, KeyError, TypeError):
            layers = []
        rates = []
        for layer in layers:
            shp = layer.get('config', {}).get('batch_input_shape')
            if shp and len(shp) >= 3:
                config['window_shape'] = (int(shp[1]), int(shp[2]))
            if layer.get('class_name') == 'Dropout':
                rates.append(float(layer['config'].get('rate', layer['config'].get('p', 0.25))))
        if rates:
            config['prop_dropout_base'] = rates[0]           # the first Dropout carries drp itself (:179)
    out = dict(weights=weights, config=config, optimizer=None, loss=None)
    tc = f.attrs.get('training_config')
    if tc is not None:
        tc = json.loads(_s(tc))
        loss = tc.get('loss')
        out['loss'] = loss if isinstance(loss, str) else None
        oc = tc.get('optimizer_config', {})
        if oc.get('class_name') == 'Adam' and 'optimizer_weights' in f:
            og = f['optimizer_weights']
            names = [_s(n) for n in np.atleast_1d(og.attrs['weight_names'])]
            vals = [og[n].read() for n in names]
            n_train = (len(vals) - 1) // 2
            if len(vals) == 2 * n_train + 1 and n_train > 0:
                out['optimizer'] = dict(config=oc.get('config', {}), iterations=int(np.asarray(vals[0]).reshape(-1)[0]),
                                        m=[np.asarray(v, np.float32) for v in vals[1:1 + n_train]],
                                        v=[np.asarray(v, np.float32) for v in vals[1 + n_train:]])
    return out


# ---- writing ------------------------------------------------------------------------------------------------------------
def write_keras_model(path, weights, config, optimizer=None, loss=None, metrics=None):
    """weights: get_weights()-ordered arrays; config: Model.config; optimizer: None or dict(config, iterations, m, v) with
    m / v lists in trainable-weight order (kernel, bias[, gamma, beta] per layer)."""
    ups = config.get('upsampling_or_transpose', 'transpose') != 'transpose'
    seq = keras_layer_sequence(config['nb_filters_base'], config.get('prop_dropout_base', 0.25), ups, config['window_shape'])
    expected = sum(len(ws) for *_, ws in seq)
    if len(weights) != expected:
        raise ValueError('expected %d weight arrays, got %d' % (expected, len(weights)))
    w = hdf5_min.Writer()
    w.attrs['keras_version'] = KERAS_VERSION
    w.attrs['backend'] = b'tensorflow'
    layers_json = [dict(name=n, class_name=cls, config=dict(cfg, name=n)) for n, cls, cfg, _ in seq]
    w.attrs['model_config'] = json.dumps(dict(class_name='Model', config=dict(name='unet2ds', layers=layers_json,
                                                                               input_layers=[[seq[0][0], 0, 0]],
                                                                               output_layers=[[seq[-1][0], 0, 0]]),
                                              dcunet=dict(config, window_shape=list(config['window_shape'])))).encode('utf8')
    g = w.create_group('model_weights')
    g.attrs['layer_names'] = np.array([n.encode('utf8') for n, *_ in seq])
    g.attrs['backend'] = b'tensorflow'
    g.attrs['keras_version'] = KERAS_VERSION
    i = 0
    trainable_names = []
    for n, cls, cfg, ws in seq:
        lg = g.create_group(n)
        names = ['%s/%s' % (n, sfx) for sfx, _ in ws]
        # an empty weight_names list is stored by h5py as an empty float64 array; any empty 1-d attribute reads back as []
        lg.attrs['weight_names'] = np.array([x.encode('utf8') for x in names]) if names else np.zeros((0,), np.float64)
        for (sfx, shp), name in zip(ws, names):
            a = np.asarray(weights[i], np.float32)
            if tuple(a.shape) != tuple(shp):
                raise ValueError('weight %d (%s): shape %r != %r' % (i, name, a.shape, shp))
            lg.create_dataset(name, a)
            if sfx in ('kernel:0', 'bias:0', 'gamma:0', 'beta:0'):
                trainable_names.append(name)
            i += 1
    if optimizer is not None:
        oc = dict(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.0)
        oc.update(optimizer.get('config', {}))
        w.attrs['training_config'] = json.dumps(dict(optimizer_config=dict(class_name='Adam', config=oc),
                                                     loss=loss or 'binary_crossentropy', metrics=list(metrics or []),
                                                     sample_weight_mode=None, loss_weights=None)).encode('utf8')
        og = w.create_group('optimizer_weights')
        if len(optimizer['m']) != len(trainable_names) or len(optimizer['v']) != len(trainable_names):
            raise ValueError('optimizer state: %d / %d moment tensors for %d trainable weights'
                             % (len(optimizer['m']), len(optimizer['v']), len(trainable_names)))
        names = ['Adam/iterations:0']
        names += ['training/Adam/Variable%s:0' % ('' if k == 0 else '_%d' % k) for k in range(2 * len(trainable_names))]
        og.attrs['weight_names'] = np.array([x.encode('utf8') for x in names])
        og.create_dataset(names[0], np.array(float(optimizer['iterations']), np.float32))     # K.variable(0.) in Keras 2.0.6
        for name, val in zip(names[1:], list(optimizer['m']) + list(optimizer['v'])):
            og.create_dataset(name, np.asarray(val, np.float32))
    w.save(path)
