#!/usr/bin/env python
"""Convert a Keras-2.0.x UNet2DS model file (.hdf5, e.g. the reference's released `unet2ds_model.hdf5`,
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:28, or a ModelCheckpoint file, :423) into the
build's .npz checkpoint, which `load_model_with_new_input_shape` / `UNet2DSummary.predict` read.

Needs only numpy + h5py (any interpreter that has them, e.g. /opt/conda/bin/python3.9 in this image):

    python scripts/keras_hdf5_to_dcunet.py unet2ds_model.hdf5 unet2ds_model.npz

Keras file layout (third-party format, restated): group `model_weights` with attribute
`layer_names`; each layer group has attribute `weight_names` naming its datasets in `layer.weights` order
(Conv2D: kernel, bias; BatchNormalization: gamma, beta, moving_mean, moving_variance; Conv2DTranspose: kernel
(2,2,Cout,Cin), bias).  Concatenated in file order that IS `model.get_weights()` -- the order of the build's
134 (or 110 with UpSampling2D) arrays.  Optimizer state is not converted (training restarts Adam's moments).
"""
import json
import sys

import numpy as np


def _s(v):
    return v.decode() if isinstance(v, bytes) else str(v)


def convert(src, dst):
    import h5py
    with h5py.File(src, 'r') as f:
        g = f['model_weights'] if 'model_weights' in f else f
        weights = []
        for lname in g.attrs['layer_names']:
            lg = g[_s(lname)]
            for wname in lg.attrs['weight_names']:
                weights.append(np.asarray(lg[_s(wname)], dtype=np.float32))
        window = None
        if 'model_config' in f.attrs:
            cfg = json.loads(_s(f.attrs['model_config']))
            for layer in cfg['config']['layers']:
                shp = layer['config'].get('batch_input_shape')
                if shp:
                    window = [int(v) for v in shp[1:3]]
                    break
    if len(weights) not in (134, 110):
        raise SystemExit('expected 134 (conv-transpose) or 110 (UpSampling2D) weight arrays, found %d' % len(weights))
    nfb = int(weights[0].shape[-1])
    if weights[0].shape[:3] != (3, 3, 1):
        raise SystemExit('first array should be the (3,3,1,nfb) kernel, got %r' % (weights[0].shape,))
    meta = dict(format='dcunet-npz-1', compiled=False,
                config=dict(window_shape=window or [512, 512], nb_filters_base=nfb, prop_dropout_base=0.25,
                            upsampling_or_transpose='transpose' if len(weights) == 134 else 'upsampling'),
                source=src)
    arrays = {'w_%03d' % i: w for i, w in enumerate(weights)}
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    with open(dst, 'wb') as fp:
        np.savez(fp, **arrays)
    print('wrote %s: %d arrays, nb_filters_base=%d, window=%r' % (dst, len(weights), nfb, meta['config']['window_shape']))


if __name__ == '__main__':
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    convert(sys.argv[1], sys.argv[2])
