"""tests/golden/nf_dataset.npz: the reference's OWN nf_load_hdf5 (/root/reference/deepcalcium/datasets/nf.py:37-150) run
under this image's conda interpreter (real h5py / libhdf5, PIL) on two tiny synthetic Neurofinder directories -- what it
stores for 16-bit TIFF frames (int16 `series/raw` with libhdf5's saturating conversion, `series/mean` accumulated IN
float16 storage frame by frame, `series/max`, `masks/raw`, `masks/max`, the `name` attribute; no masks for '.test').

    /opt/conda/bin/python3.9 tests/golden/make_nf_dataset_golden.py

The download branch is not exercised (the unzipped directories exist, nf.py:78-80).  scipy.misc.imread left scipy in 1.2:
it is supplied as what it was, PIL's reader -> ndarray.  `neurofinder` / `regional` / `requests` / `tqdm` are stubbed
(unused by this function).  The fixture holds INPUTS (frames, regions) and the datasets the reference wrote.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    os.environ['HOME'] = tempfile.mkdtemp()
    from PIL import Image
    stub('neurofinder', centers=None, shapes=None)
    stub('regional', many=lambda coords: coords, one=None)
    stub('requests')
    stub('tqdm', tqdm=lambda x, *a, **k: x)
    import scipy.misc
    scipy.misc.imread = lambda p: np.array(Image.open(p))
    sys.path.insert(0, '/root/reference')
    from deepcalcium.datasets import nf as R
    import h5py

    rs = np.random.RandomState(2017)
    root = tempfile.mkdtemp()
    out = {}
    cases = [('neurofinder.00.00', (9, 24, 20), True), ('neurofinder.01.00.test', (5, 16, 28), False)]
    for name, (n, h, w), has_masks in cases:
        os.makedirs('%s/%s/images' % (root, name))
        frames = rs.randint(0, 3000, size=(n, h, w)).astype(np.uint16)
        frames[:, 0, :4] = [65535, 40000, 32768, 32767]          # beyond int16: pins the conversion on store
        frames[:, 1, :] = (np.arange(w) % 7) * 3                 # constant over time: pins the float16 accumulation
        for i in range(n):
            Image.fromarray(frames[i]).save('%s/%s/images/image%05d.tiff' % (root, name, i))
        out['frames_' + name] = frames
        if has_masks:
            os.makedirs('%s/%s/regions' % (root, name))
            regions = []
            for k in range(4):
                y0, x0 = rs.randint(0, h - 5), rs.randint(0, w - 5)
                regions.append({'coordinates': [[int(y0 + dy), int(x0 + dx)] for dy in range(4) for dx in range(3 + k % 2)]})
            with open('%s/%s/regions/regions.json' % (root, name), 'w') as fp:
                json.dump(regions, fp)
            out['regions_json_' + name] = np.frombuffer(json.dumps(regions).encode(), np.uint8)
    paths = R.nf_load_hdf5([c[0] for c in cases], datasets_dir=root)
    for (name, _, _), p in zip(cases, paths):
        assert p == '%s/%s/dataset.hdf5' % (root, name)
        with h5py.File(p, 'r') as f:
            out['attr_name_' + name] = np.frombuffer(str(f.attrs['name']).encode(), np.uint8)
            keys = []
            f.visit(lambda k: keys.append(k) if isinstance(f[k], h5py.Dataset) else None)
            out['keys_' + name] = np.frombuffer(','.join(sorted(keys)).encode(), np.uint8)
            for k in keys:
                out['%s:%s' % (name, k)] = f[k][...]
                print(name, k, f[k].dtype, f[k].shape)
    out['neurofinder_names'] = np.frombuffer(','.join(R.NEUROFINDER_NAMES).encode(), np.uint8)
    out['url_00_00'] = np.frombuffer(R.NAME_TO_URL['neurofinder.00.00'].encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, 'nf_dataset.npz'), **out)


if __name__ == '__main__':
    main()
