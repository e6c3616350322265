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


class _Labelled(object):
    """One labelled mask, reduced to what the Neurofinder scores read: its foreground pixels (flat indices, raster order) with
    their labels, per-region pixel counts and centres (= `np.array([r.mean(0) for r in mask_to_regions(m)])` bit for bit: the
    coordinate sums are integers, exact in float64 in any order).  The ground-truth side of the validation callback is
    labelled ONCE per fit()."""
    __slots__ = ('lbl', 'n', 'idx', 'lab', 'count', 'centers')

    def __init__(self, m):
        lbl, n = ndimage.label(np.asarray(m) != 0, structure=_EIGHT)
        self.lbl, self.n = lbl, int(n)
        flat = lbl.ravel()
        self.idx = np.flatnonzero(flat)
        self.lab = flat[self.idx]
        yy, xx = np.divmod(self.idx, lbl.shape[1])
        self.count = np.bincount(self.lab, minlength=n + 1)[1:].astype(np.float64)
        sy = np.bincount(self.lab, weights=yy.astype(np.float64), minlength=n + 1)[1:]
        sx = np.bincount(self.lab, weights=xx.astype(np.float64), minlength=n + 1)[1:]
        self.centers = np.stack([sy / self.count, sx / self.count], axis=1) if n else np.zeros((0, 2))


def _match_matrix(D, threshold):
    """_match on a precomputed distance matrix D[i, j] = |a_i - b_j|: same greedy order, same first-minimum tie-break."""
    na, nb = D.shape
    out = [None] * na
    if np.isfinite(threshold):
        # only centres closer than the threshold can ever be taken (the nearest remaining one is checked against it and
        # nothing is removed otherwise): walk each row's few candidates in (distance, index) order
        ii, jj = np.nonzero(D < threshold)
        if not ii.size:
            return out
        order = np.lexsort((jj, D[ii, jj], ii))
        taken = set()
        for i, j in zip(ii[order].tolist(), jj[order].tolist()):
            if out[i] is None and j not in taken:
                out[i] = j
                taken.add(j)
        return out
    alive = np.ones(nb, dtype=bool)
    left = nb
    inf = np.inf
    for i in range(na):
        if not left:
            break
        row = np.where(alive, D[i], inf)
        j = int(np.argmin(row))
        if row[j] < threshold:
            out[i] = j
            alive[j] = False
            left -= 1
    return out


def score_labelled(a, b):
    """(precision, recall, inclusion, exclusion, F1) of prediction `b` against truth `a` (both _Labelled, same shape):
    `centers(ra, rb)` + `shapes(ra, rb)` of the region lists, computed from the label images -- centre distances as one
    matrix, region overlaps as one joint histogram over the prediction's foreground pixels instead of Python sets of
    coordinate tuples.  Bit-identical to the list versions above (tests/test_nf_matching.py)."""
    if a.n == 0 or b.n == 0:
        return 0., 0., 0., 0., 0.
    D = np.sqrt(((b.centers[None, :, :] - a.centers[:, None, :]) ** 2).sum(2))
    hits = sum(1 for j in _match_matrix(D, 5.0) if j is not None)
    r, p = hits / float(a.n), hits / float(b.n)
    m_inf = _match_matrix(D, np.inf)
    pi = [i for i, j in enumerate(m_inf) if j is not None]
    if pi:
        pj = [m_inf[i] for i in pi]
        la = a.lbl.ravel()[b.idx]                      # the truth's label under every predicted foreground pixel
        ov = np.bincount(la.astype(np.int64) * (b.n + 1) + b.lab, minlength=(a.n + 1) * (b.n + 1)).reshape(a.n + 1, b.n + 1)
        hit = ov[np.asarray(pi) + 1, np.asarray(pj) + 1].astype(np.float64)
        inc = float(np.mean(hit / a.count[pi]))
        exc = float(np.mean(hit / b.count[pj]))
    else:
        inc, exc = 0.0, 0.0
    f1 = 2. * (r * p) / (r + p) if (r + p) > 0 else 0.
    return p, r, inc, exc, f1


def nf_mask_metrics(m, mp):
    """(precision, recall, inclusion, exclusion, F1) of a predicted 2-D mask, datasets/nf.py:153-174.
    All-zero prediction -> five zeros (:165-166).  Where the reference would divide 0/0 (no matched region:
    a ZeroDivisionError there) this returns F1 = 0.  `m` may be a _Labelled (a ground truth labelled once)."""
    mp = np.asarray(mp)
    if np.sum(mp.round()) == 0:
        return 0., 0., 0., 0., 0.
    a = m if isinstance(m, _Labelled) else _Labelled(m)
    if a.n == 0:
        return 0., 0., 0., 0., 0.
    return score_labelled(a, _Labelled(mp))


def nf_mask_metrics_lists(m, mp):
    """nf_mask_metrics through the region-list restatements of neurofinder's centers / shapes above (the slow, literal form;
    kept as the cross-check of the label-image form)."""
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


class NativeScorer(object):
    """nf_mask_metrics as ONE native call (dc_host_nf_pairs, include/dcunet.h; csrc/nf_score.cpp): the same labelling,
    matching and overlap arithmetic, bit-identical results, ~30x less host time and no GIL held -- what the validation
    callback's scoring threads run.  One instance per thread (it owns the call's scratch); masks: 2-D uint8, non-zero =
    foreground, of at most `max_pixels` pixels."""

    def __init__(self, max_pixels):
        from ._lib import lib
        self._fn = lib().dc_host_nf_pairs
        self._ws_bytes = lib().dc_host_nf_ws_bytes
        self.max_pixels = int(max_pixels)
        self.cap = self.max_pixels // 2 + 2
        self.counts = np.zeros(4, np.int32)
        self.inc, self.exc = np.empty(self.cap, np.float64), np.empty(self.cap, np.float64)
        self.ws = np.empty(self._ws_bytes(1, self.max_pixels) + 64, np.uint8)     # (1 x n is the worst case of every H x W = n)

    def __call__(self, m, mp):
        a = m if (isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.flags.c_contiguous) else \
            np.ascontiguousarray(np.asarray(m) != 0, dtype=np.uint8)
        b = mp if (isinstance(mp, np.ndarray) and mp.dtype == np.uint8 and mp.flags.c_contiguous) else \
            np.ascontiguousarray(np.asarray(mp) != 0, dtype=np.uint8)
        if a.shape != b.shape or a.ndim != 2 or a.size > self.max_pixels:
            raise ValueError('masks of shapes %r and %r (scorer sized for %d pixels)' % (a.shape, b.shape, self.max_pixels))
        if a.size == 0 or not b.any():
            return 0., 0., 0., 0., 0.
        self._fn(a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1], 5.0, self.counts.ctypes.data, self.inc.ctypes.data,
                 self.exc.ctypes.data, self.cap, self.ws.ctypes.data)
        na, nb, hits, npairs = (int(v) for v in self.counts)
        if na == 0 or nb == 0:
            return 0., 0., 0., 0., 0.
        r, p = hits / float(na), hits / float(nb)
        i, e = (float(np.mean(self.inc[:npairs])), float(np.mean(self.exc[:npairs]))) if npairs else (0.0, 0.0)
        f1 = 2. * (r * p) / (r + p) if (r + p) > 0 else 0.
        return p, r, i, e, f1


def nf_mask_metrics_native(m, mp):
    """One-off form of NativeScorer (allocates its scratch per call)."""
    return NativeScorer(int(np.asarray(m).size))(m, mp)


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
