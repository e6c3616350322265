"""Generate tests/golden/*.npz by IMPORTING the reference's own host-side functions.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + expected outputs) and travel with the repo, the
reference does not.  Third-party modules the reference imports but that are not
installed here (keras, h5py, skimage, regional, neurofinder, scipy.misc) are
replaced by empty stubs: none of the captured functions calls into them, except
`keras.backend`, for which a numpy stand-in with Keras' documented semantics
(round = half-to-even, epsilon = 1e-7) is supplied so that the metric/loss
algebra of deepcalcium/utils/neurons.py:13-106 can be evaluated on arrays.

    python tests/golden/make_goldens.py
"""
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    os.environ['HOME'] = tempfile.mkdtemp()          # utils/config.py:36 creates ~/.deep-calcium at import

    class _Any(object):
        def __init__(self, *a, **k):
            pass

    K = _stub('keras.backend', round=np.round, sum=np.sum, clip=np.clip, log=np.log,
              epsilon=lambda: 1e-7, ones_like=np.ones_like, backend=lambda: 'tensorflow',
              mean=np.mean)
    _stub('keras', backend=K)
    _stub('keras.callbacks', Callback=_Any, ModelCheckpoint=_Any, CSVLogger=_Any, ReduceLROnPlateau=_Any)
    _stub('keras.optimizers', Adam=_Any)
    _stub('keras.losses', binary_crossentropy=None)
    sm = _stub('scipy.misc', imsave=None, imread=None)
    import scipy
    scipy.misc = sm
    class _FakeH5(object):
        store = {}

        def __init__(self, path, *a, **k):
            self.d = _FakeH5.store[path]
            self.attrs = {'name': self.d.get('name')}

        def get(self, key):
            return self.d[key]

        def __contains__(self, key):
            return key in self.d

        def close(self):
            pass
    _stub('h5py', File=_FakeH5)
    _stub('skimage')
    _stub('skimage.color', gray2rgb=None, rgb2gray=None)
    _stub('skimage.measure')
    _stub('skimage', measure=sys.modules['skimage.measure'], color=sys.modules['skimage.color'])
    _stub('regional', one=None, many=None)
    _stub('neurofinder', centers=None, shapes=None, match=None)
    _stub('requests')
    _stub('tqdm', tqdm=lambda x, *a, **k: x)
    sys.path.insert(0, REF)


def synth_datasets():
    """Two synthetic 64x64 'summary images' with blob masks (deterministic, independent of the global RNG)."""
    rs = np.random.RandomState(1234)
    S, M = [], []
    for k in range(2):
        s = rs.standard_normal((64, 64)).astype(np.float32)
        m = np.zeros((64, 64), np.uint8)
        for _ in range(12):
            cy, cx = rs.randint(3, 61, 2)
            m[cy - 2:cy + 3, cx - 2:cx + 3] = 1
        S.append(s)
        M.append(m)
    return S, M


def main():
    install_stubs()
    from deepcalcium.models.neurons.unet_2d_summary import UNet2DSummary, _ValidationMetricsCB
    from deepcalcium.utils import neurons as RN

    # (i) TTA table, deepcalcium/utils/neurons.py:112-137
    x = np.arange(2 * 16 * 16, dtype=np.float32).reshape(2, 16, 16)
    tta = {'x': x}
    for name, aug, inv in RN.INVERTIBLE_2D_AUGMENTATIONS:
        tta['aug_' + name] = np.ascontiguousarray(aug(x))
        tta['inv_' + name] = np.ascontiguousarray(inv(aug(x)))
        # inverse applied to a NON-symmetric "prediction" (what predict() does, unet_2d_summary.py:588-589)
        tta['invp_' + name] = np.ascontiguousarray(inv(np.ascontiguousarray(aug(x)) * 2 + 1))
    tta['names'] = np.array([n for n, _, _ in RN.INVERTIBLE_2D_AUGMENTATIONS])
    np.savez_compressed(os.path.join(OUT, 'tta.npz'), **tta)

    # (ii) _batch_gen, unet_2d_summary.py:434-530, global numpy RNG seeded as examples/neurons/unet2ds_nf.py:18
    S, M = synth_datasets()
    model = UNet2DSummary(cpdir=tempfile.mkdtemp())
    out = {'S0': S[0], 'S1': S[1], 'M0': M[0], 'M1': M[1]}
    for tag, nb_aug in (('aug15', 15), ('aug0', 0)):
        np.random.seed(865)
        gen = model._batch_gen(S, M, ['a', 'b'], [(0, 48), (0, 48)], 4, 10, (32, 32), nb_aug)
        for i in range(3):
            sb, mb = next(gen)
            out['%s_s%d' % (tag, i)] = sb.copy()
            out['%s_m%d' % (tag, i)] = mb.copy()
    # ragged: window larger than the sampled stripe -> zero-filled tail (:520-521)
    np.random.seed(865)
    gen = model._batch_gen(S, M, ['a', 'b'], [(0, 24), (0, 24)], 2, 10, (32, 32), 3)
    sb, mb = next(gen)
    out['ragged_s0'], out['ragged_m0'] = sb.copy(), mb.copy()
    np.savez_compressed(os.path.join(OUT, 'batch_gen.npz'), **out)

    # (iii) _ValidationMetricsCB.__init__, unet_2d_summary.py:34-60
    cb = _ValidationMetricsCB(None, S, M, ['a', 'b'], [(48, 64), (48, 64)])
    val = {'val_coords': np.array(cb.val_coords), 'names': np.array(cb.names)}
    for i, (s, m) in enumerate(zip(cb.S_summ, cb.M_summ)):
        val['S%d' % i] = np.ascontiguousarray(s)
        val['M%d' % i] = np.ascontiguousarray(m)
    np.savez_compressed(os.path.join(OUT, 'val_cb.npz'), **val)

    # (iv) metric / loss algebra, deepcalcium/utils/neurons.py:13-106, on arrays through the numpy K stand-in
    rs = np.random.RandomState(99)
    yt = (rs.random_sample((3, 24, 24)) < 0.2).astype(np.float64)
    yp = np.clip(yt * 0.7 + rs.random_sample((3, 24, 24)) * 0.45, 0, 1)
    yp.flat[:7] = 0.5                                  # exercise half-to-even rounding
    met = {'yt': yt, 'yp': yp}
    for fn in ('prec', 'reca', 'F1', 'jacc', 'dice', 'dicesq', 'posyt', 'posyp',
               'jacc_loss', 'dice_loss', 'dicesq_loss'):
        met[fn] = np.float64(getattr(RN, fn)(yt, yp))
    met['weighted_binary_crossentropy'] = RN.weighted_binary_crossentropy(yt, yp)
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **met)
    # (v) _summarize_series / _summarize_mask, unet_2d_summary.py:227-291, through a dict-backed h5py.File stand-in
    import h5py
    rs = np.random.RandomState(5)
    mean_img = (rs.random_sample((40, 48)) * 900 + 100).astype(np.float16)
    msk = np.zeros((6, 40, 48), np.int8)
    msk[0, 5:12, 5:12] = 1           # isolated neuron
    msk[1, 10:18, 10:18] = 1         # overlaps neuron 0
    msk[2, 20:26, 5:11] = 1
    msk[3, 20:26, 11:17] = 1         # touches neuron 2 (adjacent columns)
    msk[4, 30:36, 30:38] = 1
    msk[5, 35:39, 37:44] = 1         # overlaps + touches neuron 4 diagonally
    h5py.File.store['fake.hdf5'] = {'series/mean': mean_img, 'masks/raw': msk, 'name': 'neurofinder.99.99'}
    from deepcalcium.models.neurons import unet_2d_summary as U
    np.savez_compressed(os.path.join(OUT, 'summaries.npz'), series_mean=mean_img, masks_raw=msk,
                        name=np.array('neurofinder.99.99'),
                        summ_series=U._summarize_series('fake.hdf5'), summ_mask=U._summarize_mask('fake.hdf5'))
    # (vi) _summarize_mask on crowded random stacks (many touching / overlapping / diagonal seams): pins the
    # order-dependent sequential deletion (unet_2d_summary.py:275-284) far beyond the 6-neuron case above
    rnd = {}
    for case, (n, hh, ww, rmax, seed) in enumerate([(40, 72, 80, 6, 11), (90, 96, 96, 5, 12), (25, 48, 64, 9, 13)]):
        rs = np.random.RandomState(seed)
        msk = np.zeros((n, hh, ww), np.int8)
        gy, gx = np.mgrid[:hh, :ww]
        for z in range(n):
            cy, cx = rs.randint(0, hh), rs.randint(0, ww)
            ry, rx = rs.randint(2, rmax + 1), rs.randint(2, rmax + 1)
            msk[z] = (((gy - cy) / ry) ** 2 + ((gx - cx) / rx) ** 2 <= 1.0)
        h5py.File.store['rand%d.hdf5' % case] = {'masks/raw': msk, 'name': 'rand%d' % case}
        rnd['masks_raw_%d' % case] = np.packbits(msk.astype(np.uint8), axis=None)
        rnd['shape_%d' % case] = np.array(msk.shape)
        rnd['summ_mask_%d' % case] = U._summarize_mask('rand%d.hdf5' % case).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, 'summaries_rand.npz'), **rnd)
    print('wrote', sorted(f for f in os.listdir(OUT) if f.endswith('.npz')))


if __name__ == '__main__':
    main()
