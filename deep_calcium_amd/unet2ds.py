"""UNet2DSummary: the reference's fit()/predict() API surface over the HIP model.

Same class name, constructor, `fit`, `_batch_gen` and `predict` signatures and behaviour as
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:301-625, with `net_builder_func` defaulting to
`unet_hip` (the drop-in for `unet()`), and checkpoints in the build's own .npz container.
Host-side numpy only; every tensor op happens behind `model.predict` / `model.fit_generator`.
"""
from __future__ import division, print_function

import logging
import os
import pickle
from time import time

import numpy as np

from . import parallel
from .model import (Adam, Callback, CSVLogger, DeviceBatch, ModelCheckpoint, ReduceLROnPlateau, crop_layout, unet_hip,
                    load_model_with_new_input_shape)
from .nf_metrics import nf_mask_metrics

# ---- 8 invertible test-time augmentations on (N,H,W) batches: (name, forward, inverse) -----------------------
# Same table as deepcalcium/utils/neurons.py:112-137, written as (k rot90s, then optional flip of an axis).


def _tta(k, flip_axis):
    def fwd(x):
        y = np.rot90(x, k, axes=(1, 2)) if k else x
        return np.flip(y, flip_axis) if flip_axis is not None else y

    def inv(x):
        y = np.flip(x, flip_axis) if flip_axis is not None else x
        return np.rot90(y, -k, axes=(1, 2)) if k else y
    return fwd, inv


INVERTIBLE_2D_AUGMENTATIONS = [
    ('identity',) + _tta(0, None),
    ('vflip',) + _tta(0, 1),
    ('hflip',) + _tta(0, 2),
    ('rot90',) + _tta(1, None),
    ('rot180',) + _tta(2, None),
    ('rot270',) + _tta(3, None),
    ('rot90vflip',) + _tta(1, 1),
    ('rot90hflip',) + _tta(1, 2),
]
# NB the reference's inverse for the last two entries is the forward map applied again (rot90 then flip),
# which is its own inverse for these two compositions only on SQUARE windows; _tta's flip-then-rot(-1) is the
# exact inverse in general and coincides with it on square windows (asserted by tests/test_host.py).

# ---- the 6 training / validation augmentations (unet_2d_summary.py:54-59, :459-466) --------------------------
_SIX = [
    lambda a: a,
    lambda a: a[:, ::-1],
    lambda a: a[::-1, :],
    lambda a: np.rot90(a, 1),
    lambda a: np.rot90(a, 2),
    lambda a: np.rot90(a, 3),
]


_PROBE = np.arange(4).reshape(2, 2)
_D4 = [lambda a: a, lambda a: a[:, ::-1], lambda a: a[::-1, :], lambda a: np.rot90(a, 1), lambda a: np.rot90(a, 2),
       lambda a: np.rot90(a, 3), lambda a: np.rot90(a, 1)[:, ::-1], lambda a: np.rot90(a, 1)[::-1, :]]
_D4_KEYS = {tuple(f(_PROBE).ravel()): f for f in _D4}
_D4[0] = _SIX[0]
_D4_KEYS[tuple(_PROBE.ravel())] = _SIX[0]


def _d4_bits(f):
    """f (one of the 8 dihedral maps of a square array) as the 3 bits of dc_crop_augment (include/dcunet.h):
    f(a)[i][j] = a[r][c] with (r, c) = bit 0 ? (j, i) : (i, j), then r -> n-1-r if bit 1, c -> n-1-c if bit 2.  Found by
    applying f itself to a row-index and a column-index image: the device cannot disagree with numpy's rot90 / flips."""
    n = 3
    R, C = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
    fr, fc = np.asarray(f(R)), np.asarray(f(C))
    for bits in range(8):
        r, c = (C, R) if bits & 1 else (R, C)
        r = n - 1 - r if bits & 2 else r
        c = n - 1 - c if bits & 4 else c
        if np.array_equal(fr, r) and np.array_equal(fc, c):
            return bits
    raise ValueError('not a dihedral map')


_D4_BITS = {id(f): _d4_bits(f) for f in _D4}


def _compose_six(indices):
    """The single dihedral transform equal to applying _SIX[j] for j in `indices` in order (square arrays)."""
    probe = _PROBE
    for j in indices:
        probe = _SIX[j](probe)
    return _D4_KEYS[tuple(np.asarray(probe).ravel())]


def _open_dataset(dspath):
    """A dataset is the reference's HDF5 file (datasets/nf.py:37-150: attrs['name'], series/{raw,mean,max},
    masks/{raw,max}; read with h5py when importable, else with the built-in reader hdf5_min) or an .npz with the same
    members, '/' -> '_'."""
    if str(dspath).endswith('.npz'):
        z = np.load(dspath, allow_pickle=False)
        return {k.replace('_', '/', 1): z[k] for k in z.files}, None
    try:
        import h5py
        return None, h5py.File(dspath, 'r')
    except ImportError:
        from . import hdf5_min
        return None, hdf5_min.File(dspath)


def _get(dspath, key):
    d, fp = _open_dataset(dspath)
    if d is not None:
        return d[key]
    try:
        if key == 'name':
            return fp.attrs[key]
        node = fp[key]
        return node.read() if hasattr(node, 'read') else node[...]
    finally:
        fp.close()


def _summarize_series(dspath):
    """Normalised mean image, unet_2d_summary.py:227-241."""
    summ = np.asarray(_get(dspath, 'series/mean')).astype(np.float32)
    return (summ - np.mean(summ)) / np.std(summ)


def _summarize_mask(dspath):
    """Flatten the neuron mask stack to one (H,W) mask, dropping pixels owned by >1 neuron and pixels whose
    8-neighbourhood (among the surviving single-owner pixels) touches another neuron, unet_2d_summary.py:244-291.
    The reference removes neighbourhoods sequentially in dict order while iterating a snapshot of the keys, so the
    result depends on that order.  Surviving pixels only ever disappear, hence only pixels whose INITIAL neighbourhood
    holds two owners can ever trigger a removal: those candidates (the seams between touching neurons, found with
    array ops) are swept in the reference's key order; everything else is vectorised."""
    return _flatten_mask_stack(np.asarray(_get(dspath, 'masks/raw')))


def _flatten_mask_stack(msks):
    zz, yy, xx = np.where(msks == 1)
    H, W = msks.shape[1:]
    count = np.zeros((H, W), np.int32)
    np.add.at(count, (yy, xx), 1)
    owner = np.full((H, W), -1, np.int64)
    owner[yy, xx] = zz
    owner[count != 1] = -1
    alive = owner >= 0
    # smallest / largest owner over the 3x3 neighbourhood of surviving pixels
    big = np.iinfo(np.int64).max
    lo = np.pad(np.where(alive, owner, big), 1, constant_values=big)
    hi = np.pad(owner, 1, constant_values=-1)
    nlo, nhi = lo[1:-1, 1:-1].copy(), hi[1:-1, 1:-1].copy()
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            np.minimum(nlo, lo[dy:dy + H, dx:dx + W], out=nlo)
            np.maximum(nhi, hi[dy:dy + H, dx:dx + W], out=nhi)
    cand = (nhi >= 0) & (nlo != nhi) & (count == 1)
    # the reference's dict keys in insertion order = np.where order (z, then y, then x) of the single-owner pixels
    sel = cand[yy, xx]
    for y, x in zip(yy[sel].tolist(), xx[sel].tolist()):
        y0, y1, x0, x1 = max(y - 1, 0), min(y + 2, H), max(x - 1, 0), min(x + 2, W)
        a = alive[y0:y1, x0:x1]
        o = owner[y0:y1, x0:x1][a]
        if o.size and o.min() != o.max():
            a[...] = False
    return alive.astype(np.float64)


def _name_dataset(dspath):
    name = _get(dspath, 'name')
    if isinstance(name, np.ndarray):
        name = name.item() if name.shape == () else str(name)
    return name.decode() if isinstance(name, bytes) else str(name)


class _ValidationMetricsCB(Callback):
    """Full-size validation at epoch end, unet_2d_summary.py:31-120: copies the training weights into the
    512^2 model, predicts 6 augmented copies of every dataset, scores the validation stripe with the
    Neurofinder metrics and writes val_nf_* into the shared logs dict."""

    def __init__(self, model_val, S_summ, M_summ, names, y_coords, scores_path=None):
        super(_ValidationMetricsCB, self).__init__()
        self.model_val = model_val
        self.S_summ, self.M_summ, self.val_coords, self.names = [], [], [], []
        self.scores_path = scores_path
        for s, m, name, (y0, y1) in zip(S_summ, M_summ, names, y_coords):
            stripe = np.zeros(s.shape, dtype=np.uint8)
            stripe[y0:y1, :] = 1
            for f in _SIX:
                self.S_summ.append(f(s))
                self.M_summ.append(f(m))
                self.names.append(name)
                yy, xx = np.where(f(stripe) == 1)
                # inclusive max indices, later used as exclusive slice ends (drops one row/col) -- kept as is
                self.val_coords.append([min(yy), max(yy), min(xx), max(xx)])

    def on_epoch_end(self, epoch, logs={}):
        logger = logging.getLogger('_ValidationMetricsCB')
        tic = time()
        if parallel.world_size() > 1 and getattr(self.model, 'engine', None) is not None:
            parallel.sync_moving_stats(self.model.engine.sflat)
        eng_val = getattr(self.model_val, 'engine', None)
        if eng_val is not None:
            # Drain the device before the epoch's ~5 000 validation launches are enqueued.  Without it the host RESIDENT SET grew
            # by 0.19 GB at every epoch end of the 19-dataset run (anonymous memory of the HIP runtime, none of it Python's:
            # tracemalloc flat, pinned allocator flat; the same callback called five times in a row on an idle device grows nothing,
            # and a synchronisation AFTER it does not help) -- 4.3 GB after 25 epochs.  The last training step is the only work
            # pending here (every step already waits for its own metric sums), so the wait is at most one step per epoch:
            # 299.1 vs 297-300 steps/s (profiles/r06_soak_fit.txt).
            import torch
            torch.cuda.synchronize(eng_val.device)
        if hasattr(self.model_val, 'copy_weights_from'):
            self.model_val.copy_weights_from(self.model)          # device to device when both are HIP models on one GPU
        else:
            self.model_val.set_weights(self.model.get_weights())  # (:69)
        n = len(self.S_summ)
        if getattr(self.model_val, 'engine', None) is not None:
            scores = self._score_on_device(n)
        else:
            scores = self._score_through_predict(n)
        pp, rr, ff = scores[:, 0].tolist(), scores[:, 1].tolist(), scores[:, 2].tolist()
        name_to_f1 = {nm: [] for nm in self.names}
        for p, r, f, name in zip(pp, rr, ff, self.names):
            name_to_f1[name].append(f)
            logger.info('%s p=%.3lf r=%.3lf f=%.3lf' % (name, p, r, f))
        if self.scores_path and parallel.rank() == 0:
            with open(self.scores_path, 'wb') as fp:
                pickle.dump(name_to_f1, fp)
        eps = 1e-4 * epoch if epoch else 0
        logs['val_nf_f1_mean'] = np.mean(ff) + eps
        logs['val_nf_f1_median'] = np.median(ff) + eps
        logs['val_nf_f1_min'] = np.min(ff) + eps
        logs['val_nf_f1_adj'] = np.mean(ff) * np.min(ff) + eps
        logs['val_nf_prec'] = np.mean(pp)
        logs['val_nf_reca'] = np.mean(rr)
        logger.info('mean f1 = %.3lf  (validation %.3lf s)' % (logs['val_nf_f1_mean'], time() - tic))

    def _score_through_predict(self, n):
        """The reference's own sequence (:76-91) for a foreign Keras-shaped model: predict, round, score -- one by one."""
        _, hw, ww = self.model_val.input_shape
        out = np.zeros((n, 3))
        for i, (s, m, (y0, y1, x0, x1)) in enumerate(zip(self.S_summ, self.M_summ, self.val_coords)):
            batch = np.pad(s, ((0, hw - s.shape[0]), (0, ww - s.shape[1])), 'reflect')[np.newaxis].astype(np.float32)
            mp = self.model_val.predict(batch)[0]
            p, r, _, _, f = nf_mask_metrics(m[y0:y1, x0:x1], mp[y0:y1, x0:x1].round())
            out[i] = (p, r, f)
        return out

    # Only the validation stripe of every forward is scored (:90-91).  An output pixel of this network depends on the input within
    # 102 pixels of it (measured with the float64 oracle, worst pooling alignment; 2 + 4 + ... analytically <= 107): forwarding the
    # stripe plus a HALO of 112 rows / columns (a multiple of 16, so every pooling grid stays where it was) gives the stripe's
    # probabilities BIT FOR BIT -- every kernel's per-pixel arithmetic is position-independent and inference BatchNorm is a
    # per-channel affine -- at about half the FLOPs of the 512 x 512 forward (tests/test_fit_device_gpu.py holds the equality).
    HALO = 112

    def _crop_window(self, c, hw, ww):
        """(cy0, cy1, cx0, cx1): the part of the (hw, ww) padded image that has to be forwarded for the scored window c."""
        y0, y1, x0, x1 = (int(v) for v in c)
        R = self.HALO
        if R is None or y1 <= y0 or x1 <= x0:
            return 0, hw, 0, ww
        cy0, cy1 = max(0, (y0 - R) // 16 * 16), min(hw, -(-(y1 + R) // 16) * 16)
        cx0, cx1 = max(0, (x0 - R) // 16 * 16), min(ww, -(-(x1 + R) // 16) * 16)
        if (cy1 - cy0) * (cx1 - cx0) > 0.8 * hw * ww:
            return 0, hw, 0, ww
        return cy0, cy1, cx0, cx1

    def _score_on_device(self, n):
        """The 6 n validation forwards + scorings of one epoch (:76-91), organised for the hardware:
          * the reflect-padded images are uploaded ONCE per fit() and stay in HBM -- cropped to the scored stripe + HALO (above);
          * under data parallelism the items are dealt round-robin to the ranks (inference is 'replicas only'),
            the per-item scores are summed over the ranks -- every item is owned by exactly one rank, the others add 0.0 --,
            so every rank writes the SAME logs;
          * forwards run in batches of 8 per window shape (inference BatchNorm is per-image), the probabilities never leave the
            device: the scored stripe `mp[y0:y1, x0:x1].round()` (:91) comes back as one byte per pixel (dc_round_window_u8);
          * a scoring thread takes each batch as its copy lands and runs the native scorer (dc_host_nf_pairs: bit-identical
            to nf_mask_metrics, GIL released) while the device is on the next batch."""
        import threading
        import torch
        from .net import UNetEngine
        from .nf_metrics import NativeScorer
        eng = self.model_val.engine
        _, hw, ww = self.model_val.input_shape
        world, rank = parallel.world_size(), parallel.rank()
        mine = list(range(rank, n, world))
        st = getattr(self, '_dev', None)
        if st is None or st['key'] != (hw, ww, n, world, rank, self.HALO):
            truth = [np.ascontiguousarray(np.asarray(self.M_summ[i])[c[0]:c[1], c[2]:c[3]] != 0, dtype=np.uint8)
                     for i, c in ((i, self.val_coords[i]) for i in mine)]
            sizes = [t.size for t in truth]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            groups = {}                              # window shape -> items (positions in `mine`), their crops
            for k, i in enumerate(mine):
                win = self._crop_window(self.val_coords[i], hw, ww)
                s_pad = np.pad(self.S_summ[i], ((0, hw - self.S_summ[i].shape[0]), (0, ww - self.S_summ[i].shape[1])), 'reflect')
                g = groups.setdefault((win[1] - win[0], win[3] - win[2]), dict(ks=[], wins=[], imgs=[]))
                g['ks'].append(k)
                g['wins'].append(win)
                g['imgs'].append(np.ascontiguousarray(s_pad[win[0]:win[1], win[2]:win[3]], dtype=np.float32))
            with torch.cuda.device(eng.device):
                for shape, g in groups.items():
                    g['x'] = torch.from_numpy(np.stack(g.pop('imgs'))).to(eng.device)
                    g['eng'] = eng if shape == (hw, ww) else UNetEngine(shape, eng.nfb, eng.drp, device=eng.device, mfma=eng.mfma,
                                                                      upsampling=eng.upsampling, conv_kernel_init=None)
                out_dev = torch.empty(max(int(offs[-1]), 1), dtype=torch.uint8, device=eng.device)
                out_host = torch.empty(max(int(offs[-1]), 1), dtype=torch.uint8).pin_memory()
            st = self._dev = dict(key=(hw, ww, n, world, rank, self.HALO), groups=groups, truth=truth, offs=offs, out_dev=out_dev,
                                  out_host=out_host, scorer=NativeScorer(max(sizes + [1])))
        scores = np.zeros((n, 3))
        if mine:
            with torch.cuda.device(eng.device):
                for g in st['groups'].values():          # the cropped-window engines run on the validation model's weights
                    if g['eng'] is not eng:
                        g['eng'].pflat.copy_(eng.pflat, non_blocking=True)
                        g['eng'].sflat.copy_(eng.sflat, non_blocking=True)
                        g['eng']._packed_dirty = g['eng']._fold_dirty = True
            chunks = [(g, g['ks'][j:j + 8], g['wins'][j:j + 8], j) for g in st['groups'].values() for j in range(0, len(g['ks']), 8)]
            events, flags = [], []
            done = []
            cond = threading.Condition()

            def events_iter():
                for j in range(len(chunks)):
                    with cond:
                        while len(events) <= j:
                            cond.wait()
                    yield events[j]

            def score_chunks():
                try:
                    host = st['out_host'].numpy()
                    for ev, (_, ks, _, _) in zip(events_iter(), chunks):
                        ev.synchronize()
                        for k in ks:
                            c = self.val_coords[mine[k]]
                            mp = host[st['offs'][k]:st['offs'][k + 1]].reshape(c[1] - c[0], c[3] - c[2])
                            p, r, _, _, f = st['scorer'](st['truth'][k], mp)
                            scores[mine[k]] = (p, r, f)
                except BaseException as e:      # surfaced on the caller's thread
                    done.append(e)

            th = threading.Thread(target=score_chunks, daemon=True)
            th.start()
            try:
                self._forward_chunks(eng, st, mine, chunks, events, flags, cond)
            finally:
                with cond:                       # never leave the scorer waiting for an event that will not come
                    while len(events) < len(chunks):
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(eng.device))
                        events.append(ev)
                    cond.notify_all()
                th.join()
            if done:
                raise done[0]
            if any(float(f) != 0.0 for f in torch.stack(flags).cpu().numpy().ravel()):
                # an activation left fp16's range under the optimistic guard (never on trained weights): measured bounds, again
                for g in st['groups'].values():
                    g['eng'].infer_measured = True
                return self._score_on_device(n)
        if world > 1:
            scores = parallel.all_reduce_sum_host(scores)
        return scores

    def _forward_chunks(self, eng, st, mine, chunks, events, flags, cond):
        import torch
        with torch.cuda.device(eng.device):
            stream = torch.cuda.current_stream(eng.device)
            for g, ks, wins, j0 in chunks:
                ge = g['eng']
                hc, wc = g['x'].shape[1:]
                p = ge.forward_infer(g['x'][j0:j0 + len(ks)])
                if ge.mfma == 'f16x3' and ge.range_guard and not ge.infer_measured:
                    flags.append(ge._ovf[0:1].clone())
                else:
                    flags.append(torch.zeros(1, device=eng.device))
                for j, (k, win) in enumerate(zip(ks, wins)):
                    y0, y1, x0, x1 = (int(v) for v in self.val_coords[mine[k]])
                    o0, o1 = int(st['offs'][k]), int(st['offs'][k + 1])
                    if o1 > o0:
                        eng.L.dc_round_window_u8(p[j].data_ptr(), 1, hc, wc, y0 - win[0], y1 - win[0], x0 - win[2], x1 - win[2],
                                                 st['out_dev'].data_ptr() + o0, stream.cuda_stream)
                        st['out_host'][o0:o1].copy_(st['out_dev'][o0:o1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
                with cond:
                    events.append(ev)
                    cond.notify_all()


class _PreshardedBatches(object):
    """Iterator over this rank's slices of the global batches; `presharded` tells fit_generator not to slice again."""
    presharded = True

    def __init__(self, gen):
        self._gen = gen

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    next = __next__


class UNet2DSummary(object):
    """Same constructor as the reference (unet_2d_summary.py:316-331); `net_builder_func` is the plug-point."""

    def __init__(self, cpdir=None, dataset_name_func=_name_dataset, series_summary_func=_summarize_series,
                 mask_summary_func=_summarize_mask, net_builder_func=unet_hip):
        if cpdir is None:
            cpdir = os.path.join(os.path.expanduser('~'), '.deep-calcium', 'checkpoints', 'neurons_unet2ds')
        self.cpdir = cpdir
        self.dataset_name_func = dataset_name_func
        self.series_summary_func = series_summary_func
        self.mask_summary_func = mask_summary_func
        self.net_builder_func = net_builder_func
        os.makedirs(self.cpdir, exist_ok=True)         # (several ranks may get here at once)
        self.custom_objects = {}

    def fit(self, dataset_paths, model_path=None, proceed=False, shape_trn=(96, 96), shape_val=(512, 512),
            batch_size_trn=32, batch_size_val=1, nb_steps_trn=200, nb_epochs=20, prop_trn=0.75, prop_val=0.25,
            keras_callbacks=[], optimizer=None, loss='binary_crossentropy'):
        """unet_2d_summary.py:333-432.  Returns (history dict, model path)."""
        assert len(shape_trn) == 2 and len(shape_val) == 2
        assert shape_trn[0] == shape_trn[1] and shape_val[0] == shape_val[1]
        assert 0 < prop_trn < 1 and 0 < prop_val < 1
        assert not (proceed and not model_path)
        known = ('binary_crossentropy', 'weighted_binary_crossentropy', 'dice_loss', 'dicesq_loss')
        assert loss in known or getattr(loss, '__name__', None) in known
        if optimizer is None:
            optimizer = Adam(0.002)                       # the reference's default argument (:335)

        if model_path:
            model = load_model_with_new_input_shape(model_path, shape_trn, compile=proceed,
                                                    custom_objects=self.custom_objects)
            model_val = load_model_with_new_input_shape(model_path, shape_val, compile=False,
                                                        custom_objects=self.custom_objects)
        else:
            model = self.net_builder_func(shape_trn)
            model_val = self.net_builder_func(shape_val)
            if parallel.rank() == 0:
                model.summary()
        if not proceed:
            model.compile(optimizer=optimizer, loss=loss, metrics=['F1', 'prec', 'reca', 'dice', 'dicesq', 'posyt', 'posyp'])

        names = [self.dataset_name_func(d) for d in dataset_paths]
        S_summ = [self.series_summary_func(d) for d in dataset_paths]
        M_summ = [self.mask_summary_func(d) for d in dataset_paths]
        ycval = [(s.shape[0] - int(s.shape[0] * prop_val), s.shape[0]) for s in S_summ]
        yctrn = [(0, int(s.shape[0] * prop_trn)) for s in S_summ]
        # data parallel: every rank replays the reference's single RNG stream (rank 0's state, broadcast) but only
        # materialises its own slice of each global batch
        parallel.broadcast_numpy_rng()
        shard = (parallel.rank(), parallel.world_size())
        if getattr(model, 'engine', None) is not None and os.environ.get('DC_HOST_BATCHES', '0') != '1':
            # the HIP model: summaries / masks go to HBM once, the generator sends four longs per item (dc_crop_augment)
            model.engine.set_crop_sources(S_summ, M_summ)
            gen_trn = self._device_batch_gen(S_summ, M_summ, names, yctrn, batch_size_trn, nb_steps_trn, shape_trn, 15, shard=shard)
        else:
            gen_trn = self._batch_gen(S_summ, M_summ, names, yctrn, batch_size_trn, nb_steps_trn, shape_trn, 15, shard=shard)

        tic = int(time())
        callbacks = [_ValidationMetricsCB(model_val, S_summ, M_summ, names, ycval)]
        if parallel.rank() == 0:
            callbacks += [
                CSVLogger('%s/%d_metrics.csv' % (self.cpdir, tic)),
                ModelCheckpoint('%s/%d_model_{epoch:02d}_{val_nf_f1_mean:.3f}.hdf5' % (self.cpdir, tic), mode='max',
                                monitor='val_nf_f1_mean', save_best_only=False, verbose=1, background=True),
            ]
        callbacks += [ReduceLROnPlateau(monitor='F1', factor=0.5, patience=5, min_lr=1e-4, mode='max')]
        callbacks += list(keras_callbacks)
        trained = model.fit_generator(gen_trn, steps_per_epoch=nb_steps_trn, epochs=nb_epochs, callbacks=callbacks,
                                      verbose=1, max_queue_size=1)
        self.model, self.model_val = model, model_val
        return trained.history, '%s/model_val_nf_f1_mean.hdf5' % self.cpdir

    def _batch_gen(self, S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment=0,
                   scores_path=None, shard=None):
        """Infinite generator of (s_batch (B,h,w) f32, m_batch (B,h,w) u8): neuron-centred random crops with
        random flips / rot90s, drawing from numpy's GLOBAL RNG in the reference's order
        (unet_2d_summary.py:434-530) so that a seeded run yields the reference's batches.
        shard=(r, G) (data parallel, not in the reference): EVERY random draw of the global batch is still made, in
        order -- the stream stays the single-device one -- but only items [r*B/G, (r+1)*B/G) are cropped and augmented,
        and the generator yields that slice: the array work per rank is 1/G of the global batch."""
        if shard is not None and shard[1] > 1:
            return _PreshardedBatches(self._batch_gen_impl(S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape,
                                                           nb_max_augment, scores_path, parallel.shard_slice(batch_size, *shard)))
        return self._batch_gen_impl(S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment,
                                    scores_path, None)

    def _item_stream(self, S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment,
                     scores_path, mine):
        """Phase 1 of every batch -- the reference's random stream itself (unet_2d_summary.py:479-527): EVERY draw of the
        global batch is made, in order, from numpy's global RNG; yields the items of this rank's slice `mine` (None: all) as
        (index in the slice, dataset, y0, y1, x0, x1, augmentation indices).  What is done with them -- numpy crops on the
        host (_batch_gen_impl) or four longs per item for dc_crop_augment (_device_batch_gen) -- is phase 2."""
        rng = np.random
        b0, b1 = (mine.start, mine.stop) if mine is not None else (0, batch_size)
        hw, ww = window_shape
        nb_yields = 0
        n_ds = len(S_summ)
        locs = []
        for m, (ymin, ymax) in zip(M_summ, y_coords):
            ys, xs = np.where(m[ymin:ymax, :] == 1)          # NB relative to ymin, used as absolute (:474, :510)
            locs.append(np.stack([ys, xs], axis=1))
        probs = np.ones(n_ds) / n_ds
        while True:
            if scores_path and os.path.exists(scores_path) and (nb_yields - 1) % nb_steps == 0:
                with open(scores_path, 'rb') as fp:
                    scores = pickle.load(fp)
                probs = np.array([1 - np.mean(scores[n]) for n in names])
                probs /= probs.sum()
            specs = []
            # rng.choice(np.arange(n), p=probs) == searchsorted(cumsum(p) / cumsum(p)[-1], random_sample(), 'right') and
            # rng.choice(6, n) == randint(0, 6, size=n): numpy's own (legacy RandomState) implementation, minus its
            # per-call argument checking -- same draws, same values (pinned by the reference-generated goldens)
            cdf = probs.cumsum()
            cdf /= cdf[-1]
            for b in range(batch_size):
                k = int(cdf.searchsorted(rng.random_sample(), side='right'))
                hs, ws = S_summ[k].shape
                ymin, ymax = y_coords[k]
                cy, cx = locs[k][rng.randint(0, len(locs[k]))]
                cy = min(max(ymin, cy + rng.randint(-5, 5)), ymax)
                cx = min(max(0, cx + rng.randint(-5, 5)), ws)
                y0 = max(ymin, int(cy - (hw / 2)))
                y1 = min(y0 + hw, ymax)
                x0 = max(0, int(cx - (ww / 2)))
                x1 = min(x0 + ww, ws)
                augs = rng.randint(0, len(_SIX), size=rng.randint(0, nb_max_augment + 1))
                if b0 <= b < b1:
                    specs.append((b - b0, k, y0, y1, x0, x1, augs))
            nb_yields += 1
            yield specs

    def _batch_gen_impl(self, S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment,
                        scores_path, mine):
        b0, b1 = (mine.start, mine.stop) if mine is not None else (0, batch_size)
        hw, ww = window_shape
        # crop sources in the batch dtypes (the reference's assignment into the float32 / uint8 batch arrays converts
        # per item; the values are the same)
        S_src = [np.ascontiguousarray(v, dtype=np.float32) for v in S_summ]
        M_src = [np.ascontiguousarray(v).astype(np.uint8) for v in M_summ]
        pool = None
        if (b1 - b0) * hw * ww >= 8 * 256 * 256:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=min(4, b1 - b0))
        # Phase 1 (sequential: it IS the reference's random stream) comes from _item_stream.  Phase 2: the array work of this
        # rank's items -- on a few threads for big windows (numpy's copies release the GIL).
        for specs in self._item_stream(S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment,
                                       scores_path, mine):
            s_batch = np.empty((b1 - b0, hw, ww), dtype=np.float32)
            m_batch = np.empty((b1 - b0, hw, ww), dtype=np.uint8)

            def materialise(spec):
                i, k, y0, y1, x0, x1, augs = spec
                # the drawn flips / rot90s are elements of the dihedral group: their composition is ONE of 8 pixel
                # permutations, found on a 2x2 probe and applied once (same bits, ~1/8 of the array traffic of
                # applying up to 15 of them one after the other as the reference does, :524-527)
                f = _compose_six(augs) if hw == ww else None
                if f is not None and y1 - y0 == hw and x1 - x0 == ww:  # full window: transform the crop view, one copy
                    s_batch[i], m_batch[i] = f(S_src[k][y0:y1, x0:x1]), f(M_src[k][y0:y1, x0:x1])
                    return
                s_batch[i], m_batch[i] = 0, 0                         # short crops are zero-filled (:520-521)
                m_batch[i, :y1 - y0, :x1 - x0] = M_src[k][y0:y1, x0:x1]
                s_batch[i, :y1 - y0, :x1 - x0] = S_src[k][y0:y1, x0:x1]
                if f is None:
                    for j in augs:
                        s_batch[i], m_batch[i] = _SIX[j](s_batch[i]), _SIX[j](m_batch[i])
                elif f is not _SIX[0]:
                    s_batch[i], m_batch[i] = f(s_batch[i]), f(m_batch[i])

            if pool is not None:
                list(pool.map(materialise, specs))
            else:
                for spec in specs:
                    materialise(spec)
            yield s_batch, m_batch

    def _device_batch_gen(self, S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment=0,
                          scores_path=None, shard=None):
        """_batch_gen for the HIP model's fit(): the same random stream, but every item leaves the host as FOUR longs
        (crop origin, row stride, extents, one of 8 dihedral maps) -- `dc_crop_augment` (include/dcunet.h) cuts, zero-fills
        and permutes it out of the summaries / masks resident in HBM (UNetEngine.set_crop_sources), straight into the step's
        input buffers.  No pixel leaves the host per step; the batch is bit-equal to _batch_gen's (tests/test_api_gpu.py).
        Square windows only (fit() asserts them, :365-366)."""
        hw, ww = window_shape
        assert hw == ww
        offs = crop_layout([np.shape(v) for v in S_summ])[0]
        mine = parallel.shard_slice(batch_size, *shard) if shard is not None and shard[1] > 1 else None
        n = (mine.stop - mine.start) if mine is not None else batch_size
        for specs in self._item_stream(S_summ, M_summ, names, y_coords, batch_size, nb_steps, window_shape, nb_max_augment,
                                       scores_path, mine):
            items = np.zeros((n, 4), dtype=np.int64)
            for i, k, y0, y1, x0, x1, augs in specs:
                wk = int(np.shape(S_summ[k])[1])
                items[i] = (offs[k] + y0 * wk + x0, wk, (max(y1 - y0, 0) << 32) | max(x1 - x0, 0), _D4_BITS[id(_compose_six(augs))])
            yield DeviceBatch(items, hw)

    _PREDICT_CACHE_MAX = 2

    def _predict_model(self, model_path, window_shape):
        """The inference model of predict(): the reference re-reads the model file on every call (unet_2d_summary.py:560-561);
        here the loaded model -- parsed weights, device buffers, folded BatchNorm, the resident TTA gather maps -- is kept per
        (file identity, window) and reused while the file is unchanged (same path, size and modification time).  predict() never
        changes a model's weights, so a cached one IS what a fresh load would give."""
        st = os.stat(model_path)
        key = (os.path.realpath(model_path), st.st_mtime_ns, st.st_size, tuple(window_shape))
        cache = self.__dict__.setdefault('_predict_models', [])
        for i, (k, m) in enumerate(cache):
            if k == key:
                cache.append(cache.pop(i))            # most recently used last
                return m
        model = load_model_with_new_input_shape(model_path, window_shape, compile=False, custom_objects=self.custom_objects)
        cache.append((key, model))
        del cache[:-self._PREDICT_CACHE_MAX]
        return model

    def predict(self, dataset_paths, model_path, window_shape=(512, 512), print_scores=False, save=False,
                augmentation=False, threshold=0.5):
        """unet_2d_summary.py:532-625.  Returns (Mp: list of uint8 masks, names)."""
        logger = logging.getLogger('UNet2DSummary.predict')
        model = self._predict_model(model_path, window_shape)
        assert tuple(window_shape) == (512, 512), 'TODO: implement variable window sizes.'   # as the reference (:565)
        _, hw, ww = model.input_shape
        Mp, names = [], []
        mean_prec = mean_reca = mean_comb = 0.
        # Every dataset's forward is ENQUEUED (no host synchronisation per dataset: the summary of dataset i+1 is read
        # and padded while the device works on i); with augmentation the 8 copies are made on the device, go through
        # ONE batch-8 forward (inference BatchNorm is per-image) and are inverse-mapped, averaged (float64, table
        # order), cropped and thresholded there -- without it the same path runs with the identity map alone.
        job = model.engine.tta_begin(len(dataset_paths), INVERTIBLE_2D_AUGMENTATIONS if augmentation else None)
        for i, dsp in enumerate(dataset_paths):
            names.append(self.dataset_name_func(dsp))
            s = self.series_summary_func(dsp)
            hs, ws = s.shape
            s_pad = np.pad(s, ((0, hw - hs), (0, ww - ws)), mode='reflect')
            # (:594 compares float32 probabilities with the threshold in float32 when there is no float64 TTA mean)
            job.enqueue(i, s_pad.astype(np.float32), hs, ws, threshold if augmentation else float(np.float32(threshold)))
        Mp = job.finish()
        for dsp, name, mp in zip(dataset_paths, names, Mp):
            if print_scores:
                m = self.mask_summary_func(dsp)
                prec, reca, incl, excl, comb = nf_mask_metrics(m, mp.round())
                logger.info('%s: prec=%.3lf, reca=%.3lf, incl=%.3lf, excl=%.3lf, comb=%.3lf'
                            % (name, prec, reca, incl, excl, comb))
                mean_prec += prec / len(dataset_paths)
                mean_reca += reca / len(dataset_paths)
                mean_comb += comb / len(dataset_paths)
            if save:
                # the reference renders red/blue outline PNGs (skimage/regional, out of scope); keep the mask itself
                np.save('%s/%s_mp.npy' % (self.cpdir, name), mp)
        if print_scores:
            logger.info('Mean prec=%.3lf, reca=%.3lf, comb=%.3lf' % (mean_prec, mean_reca, mean_comb))
        return Mp, names
