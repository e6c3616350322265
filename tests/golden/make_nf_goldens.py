"""tests/golden/nf.npz + nf_submission.json: the reference's OWN nf_submit and _mask_to_regional
(/root/reference/deepcalcium/datasets/nf.py:177-229) run under this image's conda interpreter, which has the real
scikit-image (skimage.measure.label does the work in both):

    /opt/conda/bin/python3.9 tests/golden/make_nf_goldens.py

`neurofinder` and `regional` are NOT installed anywhere in this image: they are stubbed (`regional.many` = identity, so
_mask_to_regional returns its plain coordinate lists; neurofinder.centers / shapes are never called here) -- the
connected-component labelling, region order, the submission's `range(1, max)` and empty-mask quirks are pinned by these
fixtures; the centre-matching scores of neurofinder==1.1.1 stay unpinned.
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
    stub('neurofinder', centers=None, shapes=None)
    stub('regional', many=lambda coords: coords, one=None)
    stub('requests')
    stub('tqdm', tqdm=lambda x, *a, **k: x)
    import scipy.misc                       # the real module (scikit-image needs it); imread left scipy in 1.2
    if not hasattr(scipy.misc, 'imread'):
        scipy.misc.imread = None
    sys.path.insert(0, '/root/reference')
    from deepcalcium.datasets import nf as R

    rs = np.random.RandomState(77)
    masks, names = [], []
    # blobs that touch diagonally (8- vs 4-connectivity), a mask with ONE component, an empty mask, a dense random mask
    m = np.zeros((40, 48), np.uint8)
    m[2:6, 2:6] = 1; m[6:9, 6:9] = 1            # diagonal touch -> one component under full connectivity
    m[20:24, 30:35] = 1; m[30:32, 5:9] = 1; m[10, 40] = 1
    masks.append(m); names.append('neurofinder.00.00.test')
    one = np.zeros((16, 16), np.uint8); one[4:9, 5:11] = 1
    masks.append(one); names.append('neurofinder.01.01')
    masks.append(np.zeros((12, 20), np.uint8)); names.append('custom_name')
    masks.append((rs.random_sample((64, 64)) < 0.18).astype(np.uint8)); names.append('neurofinder.04.01.test')
    out = {}
    for i, mk in enumerate(masks):
        out['mask_%d' % i] = mk
        if mk.max() > 0:
            regions = R._mask_to_regional(mk)
            out['nregions_%d' % i] = np.int64(len(regions))
            out['regions_%d' % i] = np.array([[k, y, x] for k, reg in enumerate(regions) for y, x in reg], np.int64)
    jp = os.path.join(OUT, 'nf_submission.json')
    R.nf_submit(masks, names, jp)
    json.load(open(jp))
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'nf.npz'), **out)
    print('wrote nf.npz, nf_submission.json:', [int(out.get('nregions_%d' % i, 0)) for i in range(len(masks))])


if __name__ == '__main__':
    main()
