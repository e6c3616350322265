"""Neurofinder mask scoring and submission files, used by the validation callback, predict(print_scores=True) and the
evaluate script.

Restates `nf_mask_metrics`, `nf_submit` and `_mask_to_regional` of /root/reference/deepcalcium/datasets/nf.py:153-229.
Pinned by fixtures made from the reference's own code run under this image's conda interpreter (real scikit-image;
tests/golden/make_nf_goldens.py): the connected-component labelling (skimage.measure.label, full connectivity, raster
label order), the region coordinate lists, and nf_submit's output byte for byte, quirks included.  NOT pinned: the
centre-matching scores -- `neurofinder==1.1.1` (centers / shapes / match) and `regional` are absent from the reference
tree and from this image, their published algorithms are restated here.
Host-side numpy/scipy: the caller of the GPU path, not part of it.
"""
import json
import logging

import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=int)      # skimage.measure.label default on 2-D input = full (8-) connectivity


def mask_to_regions(m):
    """Connected components (8-connected, raster label order) -> list of (k,2) [y,x] coordinate arrays."""
    lbl, n = ndimage.label(np.asarray(m) != 0, structure=_EIGHT)
    if n == 0:
        return []
    order = np.argsort(lbl, axis=None, kind='stable')
    flat = lbl.ravel()[order]
    start = np.searchsorted(flat, np.arange(1, n + 2))
    yy, xx = np.unravel_index(order, lbl.shape)
    return [np.stack([yy[start[i]:start[i + 1]], xx[start[i]:start[i + 1]]], axis=1) for i in range(n)]


def _match(a_centers, b_centers, threshold):
    """neurofinder.match: greedy, in order of `a`, nearest remaining centre of `b` if closer than threshold."""
    remaining = list(range(len(b_centers)))
    out = []
    for c in a_centers:
        if not remaining:
            out.append(None)
            continue
        d = np.sqrt(((b_centers[remaining] - c) ** 2).sum(1))
        j = int(np.argmin(d))
        if d[j] < threshold:
            out.append(remaining.pop(j))
        else:
            out.append(None)
    return out


def centers(a, b, threshold=5.0):
    """neurofinder.centers -> (recall, precision): matched pairs whose centre distance < threshold."""
    ca = np.array([r.mean(0) for r in a]).reshape(-1, 2)
    cb = np.array([r.mean(0) for r in b]).reshape(-1, 2)
    inds = _match(ca, cb, threshold)
    hits = 0
    for i, j in enumerate(inds):
        if j is not None and np.sqrt(((ca[i] - cb[j]) ** 2).sum()) < threshold:
            hits += 1
    return hits / float(len(a)), hits / float(len(b))


def shapes(a, b, threshold=np.inf):
    """neurofinder.shapes -> (inclusion, exclusion): mean per matched pair of |A&B|/|A| and |A&B|/|B|."""
    ca = np.array([r.mean(0) for r in a]).reshape(-1, 2)
    cb = np.array([r.mean(0) for r in b]).reshape(-1, 2)
    inds = _match(ca, cb, threshold)
    inc, exc = [], []
    for i, j in enumerate(inds):
        if j is None:
            continue
        sa = set(map(tuple, a[i].tolist()))
        sb = set(map(tuple, b[j].tolist()))
        hit = float(len(sa & sb))
        inc.append(hit / len(sa))
        exc.append(hit / len(sb))
    if not inc:
        return 0.0, 0.0
    return float(np.mean(inc)), float(np.mean(exc))


def nf_mask_metrics(m, mp):
    """(precision, recall, inclusion, exclusion, F1) of a predicted 2-D mask, datasets/nf.py:153-174.
    All-zero prediction -> five zeros (:165-166).  Where the reference would divide 0/0 (no matched region:
    a ZeroDivisionError there) this returns F1 = 0."""
    mp = np.asarray(mp)
    if np.sum(mp.round()) == 0:
        return 0., 0., 0., 0., 0.
    ra, rb = mask_to_regions(m), mask_to_regions(mp)
    if not ra:
        return 0., 0., 0., 0., 0.
    r, p = centers(ra, rb)
    i, e = shapes(ra, rb)
    f1 = 2. * (r * p) / (r + p) if (r + p) > 0 else 0.
    return p, r, i, e, f1


def nf_submit(Mp, names, json_path):
    """Neurofinder submission file for predicted masks, datasets/nf.py:177-218 -- reproduced as it is, quirks included:
    the 'neurofinder.' prefix is stripped from dataset names; coordinates are [row, col] pairs in raster order per
    region; an empty mask yields ONE dummy region [[0, 0]] (:201-202); `range(1, max)` (:205) drops the LAST labelled
    component (so a mask with a single component yields no region at all)."""
    logger = logging.getLogger('nf_submit')
    submission = []
    for mp, name in zip(Mp, names):
        if name.startswith('neurofinder.'):
            name = '.'.join(name.split('.')[1:])
        regions_all = mask_to_regions(mp)
        if not regions_all:
            regions = [{'coordinates': [[[0, 0]]]}]
        else:
            regions = [{'coordinates': [[int(y), int(x)] for y, x in r]} for r in regions_all[:-1]]
        submission.append({'dataset': name, 'regions': regions})
    with open(json_path, 'w') as fp:
        json.dump(submission, fp)
    logger.info('Saved submission to %s.' % json_path)
