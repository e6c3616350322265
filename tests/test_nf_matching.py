"""The centre / shape matching behind nf_mask_metrics (/root/reference/deepcalcium/datasets/nf.py:153-174 ->
neurofinder==1.1.1 `centers`, `shapes`, `match`; regional `overlap(method='rates')`): the package is absent from the
reference tree and from this image, so the product's restatement (deep_calcium_amd/nf_metrics.py) cannot be pinned by
running it.  Second line of defence: an INDEPENDENT restatement written here straight from the published algorithm in
the package's own shape (a shrinking target list, NaN for "no match", nanmean), compared on random region sets, plus the
properties the algorithm must have -- including that it is GREEDY in ground-truth order, not an optimal assignment."""
import math

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from deep_calcium_amd import nf_metrics as nm


# ---- independent restatement (pure Python, mirrors neurofinder/main.py's control flow) -----------------------------------
def _center(r):
    return (sum(p[0] for p in r) / float(len(r)), sum(p[1] for p in r) / float(len(r)))


def ref_match(a, b, threshold=float('inf')):
    targets = [_center(r) for r in b]
    target_inds = list(range(len(targets)))
    matches = []
    for s in a:
        c = _center(s)
        update = 1
        if len(targets) == 0:
            update = 0
        else:
            dists = [math.hypot(t[0] - c[0], t[1] - c[1]) for t in targets]
            if min(dists) < threshold:
                ind = dists.index(min(dists))            # argmin: first minimum
            else:
                update = 0
        if update == 1:
            matches.append(target_inds[ind])
            del targets[ind]
            del target_inds[ind]
        else:
            matches.append(float('nan'))
    return matches


def ref_centers(a, b, threshold=5):
    inds = ref_match(a, b, threshold)
    d = []
    for jj, ii in enumerate(inds):
        if ii == ii:                                      # not NaN
            ca, cb = _center(a[jj]), _center(b[ii])
            d.append(math.hypot(ca[0] - cb[0], ca[1] - cb[1]))
        else:
            d.append(float('inf'))
    hits = sum(1 for v in d if v < threshold)
    return hits / float(len(a)), hits / float(len(b))


def ref_shapes(a, b, threshold=float('inf')):
    inds = ref_match(a, b, threshold)
    inc, exc = [], []
    for jj, ii in enumerate(inds):
        if ii == ii:
            sa, sb = [tuple(p) for p in a[jj]], set(tuple(p) for p in b[ii])
            nhit = float(len([p for p in sa if p in sb]))
            inc.append(nhit / len(sa))                    # regional overlap 'rates': (recall, precision) of b against a
            exc.append(nhit / len(b[ii]))
    if not inc:
        return 0.0, 0.0
    return sum(inc) / len(inc), sum(exc) / len(exc)


# ---- random region sets -----------------------------------------------------------------------------------------------------
def random_mask_pair(rs, hw=96, n=14, jitter=4.0, drop=0.2, extra=3):
    """Ground truth: n discs; prediction: the same discs shifted by up to `jitter` px, some dropped, some extra."""
    m, mp = np.zeros((hw, hw), np.uint8), np.zeros((hw, hw), np.uint8)
    yy, xx = np.mgrid[:hw, :hw]
    for _ in range(n):
        cy, cx, r = rs.uniform(6, hw - 6), rs.uniform(6, hw - 6), rs.uniform(1.5, 4.0)
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
        if rs.random_sample() > drop:
            dy, dx = rs.uniform(-jitter, jitter, 2)
            mp[(yy - cy - dy) ** 2 + (xx - cx - dx) ** 2 <= (r * rs.uniform(0.7, 1.3)) ** 2] = 1
    for _ in range(extra):
        cy, cx = rs.uniform(6, hw - 6, 2)
        mp[(yy - cy) ** 2 + (xx - cx) ** 2 <= 4] = 1
    return m, mp


@pytest.mark.parametrize('seed', range(25))
def test_product_matching_equals_independent_restatement(seed):
    rs = np.random.RandomState(seed)
    m, mp = random_mask_pair(rs, n=int(rs.randint(3, 30)), jitter=float(rs.uniform(0, 7)))
    if mp.sum() == 0 or m.sum() == 0:
        pytest.skip('empty mask')
    ra, rb = nm.mask_to_regions(m), nm.mask_to_regions(mp)
    la, lb = [r.tolist() for r in ra], [r.tolist() for r in rb]
    want_m = ref_match(la, lb, 5)
    got_m = nm._match(np.array([r.mean(0) for r in ra]).reshape(-1, 2), np.array([r.mean(0) for r in rb]).reshape(-1, 2), 5)
    assert [None if w != w else w for w in want_m] == got_m
    r_, p_ = ref_centers(la, lb)
    i_, e_ = ref_shapes(la, lb)
    assert nm.centers(ra, rb) == pytest.approx((r_, p_), abs=1e-12)
    assert nm.shapes(ra, rb) == pytest.approx((i_, e_), abs=1e-12)
    p, r, i, e, f1 = nm.nf_mask_metrics(m, mp)
    assert (p, r, i, e) == pytest.approx((p_, r_, i_, e_), abs=1e-12)
    assert f1 == pytest.approx(2 * r_ * p_ / (r_ + p_) if r_ + p_ > 0 else 0.0, abs=1e-12)
    # the same number of hits sits behind both rates
    assert r * len(ra) == pytest.approx(p * len(rb), abs=1e-9)


def test_matching_properties():
    rs = np.random.RandomState(3)
    m, _ = random_mask_pair(rs, n=12)
    # identity: every region matches itself
    assert nm.nf_mask_metrics(m, m) == (1.0, 1.0, 1.0, 1.0, 1.0)
    # empty prediction: five zeros (nf.py:165-166)
    assert nm.nf_mask_metrics(m, np.zeros_like(m)) == (0., 0., 0., 0., 0.)
    # a shift of 3 px keeps every centre within the 5 px threshold, a shift of 6 px loses them all (isolated discs)
    far = np.zeros((96, 96), np.uint8)
    far[10:15, 10:15] = 1
    far[60:65, 70:75] = 1
    p, r, i, e, f1 = nm.nf_mask_metrics(far, np.roll(far, 3, axis=1))
    assert (p, r, f1) == (1.0, 1.0, 1.0) and 0 < i < 1 and i == e
    p, r, i, e, f1 = nm.nf_mask_metrics(far, np.roll(far, 6, axis=1))
    assert (p, r, f1) == (0.0, 0.0, 0.0)
    # shapes() matches with an INFINITE threshold: the far-shifted discs still pair up, with zero overlap
    assert (i, e) == (0.0, 0.0)


def test_matching_is_greedy_in_ground_truth_order_not_optimal():
    """Ground truth A (x=10), B (x=16) and predictions P2 (x=6), P1 (x=13) on one row.  Distances: A-P1 3, A-P2 4, B-P1 3,
    B-P2 10.  The optimal assignment (Hungarian on "closer than 5 px") pairs A-P2 and B-P1: two hits.  The package's pass is
    greedy in ground-truth (raster label) order: A takes its nearest prediction P1, B is left with P2 at 10 px: ONE hit.
    nf_mask_metrics must give the greedy answer -- it is what the reference's val_nf_f1 / checkpoint names are made of."""
    m, mp = np.zeros((5, 24), np.uint8), np.zeros((5, 24), np.uint8)
    m[2, 10] = m[2, 16] = 1
    mp[2, 6] = mp[2, 13] = 1
    ra, rb = nm.mask_to_regions(m), nm.mask_to_regions(mp)
    ca = np.array([r.mean(0) for r in ra])
    cb = np.array([r.mean(0) for r in rb])
    D = np.sqrt(((ca[:, None] - cb[None]) ** 2).sum(-1))
    assert D.tolist() == [[4.0, 3.0], [10.0, 3.0]]
    rows, cols = linear_sum_assignment(np.where(D < 5, 0.0, 1.0))
    assert int((D[rows, cols] < 5).sum()) == 2                      # what an optimal matcher would score
    assert nm._match(ca, cb, 5) == [1, None] == [None if w != w else w for w in ref_match([r.tolist() for r in ra], [r.tolist() for r in rb], 5)]
    p, r, i, e, f1 = nm.nf_mask_metrics(m, mp)
    assert (p, r, f1) == (0.5, 0.5, 0.5)
    # shapes() runs the same greedy pass with an infinite threshold: A-P1 and then B-P2 pair up, both without overlap
    assert (i, e) == (0.0, 0.0)


def _blob_mask(rs, h, w, n):
    m = np.zeros((h, w))
    yy, xx = np.ogrid[:h, :w]
    for _ in range(n):
        cy, cx, r = rs.randint(0, h), rs.randint(0, w), rs.randint(2, 6)
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
    return m


def test_native_scorer_is_bit_identical_to_the_list_restatement(dclib):
    """dc_host_nf_pairs (csrc/nf_score.cpp: what the validation callback's scoring thread runs) against nf_metrics' literal
    region-list restatement of neurofinder.centers / shapes AND against its label-image form, on 300 random mask pairs:
    shifted / eroded blobs, extra detections, salt noise (snake-like components, label merges), checkerboards and stripes
    (the worst cases of the union-find), empty masks on either side, 1-pixel-high images.  Equal tuples, not close ones."""
    from scipy import ndimage
    from deep_calcium_amd import nf_metrics as nm
    rs = np.random.RandomState(0)
    sc = nm.NativeScorer(130 * 520)
    lab, n = np.zeros(130 * 520, np.int32), np.zeros(1, np.int32)
    for t in range(300):
        h, w = rs.randint(1, 130), rs.randint(1, 520)
        m = _blob_mask(rs, h, w, rs.randint(0, 70))
        mp = np.roll(m, rs.randint(-4, 5, 2), (0, 1))
        mp[rs.random_sample(mp.shape) < 0.02] = 0
        if t % 3 == 0:
            mp = np.maximum(mp, _blob_mask(rs, h, w, rs.randint(0, 30)))
        if t % 5 == 0:
            mp = (rs.random_sample(mp.shape) < 0.3).astype(float)
        if t % 7 == 0:
            mp = (np.indices(mp.shape).sum(0) % 2 == 0).astype(float) if t % 14 == 0 else (np.indices(mp.shape)[1] % 2 == 0).astype(float)
        if t % 17 == 0:
            mp[:] = 0
        if t % 19 == 0:
            m[:] = 0
        want = nm.nf_mask_metrics_lists(m, mp)
        assert sc(m, mp) == want, t
        assert nm.nf_mask_metrics(m, mp) == want, t
        u = np.ascontiguousarray(mp != 0, dtype=np.uint8)
        dclib.dc_host_label8(u.ctypes.data, h, w, lab.ctypes.data, n.ctypes.data, sc.ws.ctypes.data)
        ref, nr = ndimage.label(u, structure=nm._EIGHT)
        assert nr == n[0] and np.array_equal(ref.ravel(), lab[:h * w]), t
    assert nm.nf_mask_metrics_native(m, mp) == nm.nf_mask_metrics_lists(m, mp)
