"""Keras-`Model`-shaped front end over the HIP UNet2DS engine.

This is the object the reference's plug-point expects: `UNet2DSummary(net_builder_func=...)` calls
`net_builder_func(window_shape)` (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:316-318,
:392-393) and then drives the result only through the Keras duck-type listed in SURVEY 8(b):
summary / compile / fit_generator / predict / input_shape / get_weights / set_weights / save, callbacks with a
shared mutable `logs` dict, `model.optimizer.lr`.  `unet_hip` has the signature of the reference's `unet()`
(:123-124).  Behaviour restated from Keras 2.0.6 (un-vendored).
"""
from __future__ import division, print_function

import csv
import json
import os
import queue
import threading
import time

import numpy as np
import torch

from .net import UNetEngine
from . import parallel

K_EPS = 1e-7
METRIC_NAMES = ['F1', 'prec', 'reca', 'dice', 'dicesq', 'posyt', 'posyp']


LOSS_KINDS = {'binary_crossentropy': 0, 'weighted_binary_crossentropy': 1, 'dice_loss': 2, 'dicesq_loss': 3}


def metrics_from_sums(s, count, loss='binary_crossentropy'):
    """The loss + 7 compile() metrics (deepcalcium/utils/neurons.py:13-106) from the head kernel's sums
    {bce, tp, sum round(p), fn, sum y, sum y*p, sum p^2, sum y^2[, sum p, weighted-bce]} over `count` pixels."""
    s = [float(v) for v in s]
    bce, tp, spr, fn, sy, syp, sp2, sy2 = s[:8]
    sp, wbce = (s[8], s[9]) if len(s) > 9 else (0.0, 0.0)
    prec = tp / (spr + K_EPS)
    reca = tp / (tp + fn + K_EPS)
    loss_value = {'binary_crossentropy': bce / count,
                  'weighted_binary_crossentropy': wbce / count,
                  'dice_loss': 1.0 - 2.0 * syp / (sy + sp + 1e-7),
                  'dicesq_loss': -2.0 * syp / (sy2 + sp2 + K_EPS)}[loss]
    return {
        'loss': loss_value,
        'F1': 2 * prec * reca / (prec + reca + K_EPS),
        'prec': prec,
        'reca': reca,
        'dice': 2 * tp / (sy + spr + 1e-7),
        'dicesq': 2 * syp / (sy2 + sp2 + K_EPS),
        'posyt': sy / (count + K_EPS),
        'posyp': spr / (count + K_EPS),
    }


class Adam(object):
    """keras.optimizers.Adam(lr) state holder; the update itself is dc_adam_step_flat (Keras-2.0.6 form)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.0):
        if decay:
            raise NotImplementedError('Adam(decay != 0) is not used by the reference path')
        self.lr, self.beta_1, self.beta_2, self.epsilon = float(lr), beta_1, beta_2, epsilon

    def get_config(self):
        return dict(lr=self.lr, beta_1=self.beta_1, beta_2=self.beta_2, epsilon=self.epsilon)


# ---- device-side training batches ---------------------------------------------------------------------------
def crop_layout(shapes):
    """Element offsets of the datasets' (H, W) images inside the ONE concatenated source buffer dc_crop_augment reads
    (summaries float32, masks uint8: same layout), each image starting on a 16-element boundary.  -> (offsets, total)."""
    offs, o = [], 0
    for h, w in shapes:
        offs.append(o)
        o += (int(h) * int(w) + 15) // 16 * 16
    return offs, max(o, 16)


class DeviceBatch(object):
    """One training batch as crop / augmentation items instead of pixel arrays: `items` int64 (B, 4) as dc_crop_augment
    takes them (include/dcunet.h), `window` the side of the square window.  Model.fit_generator hands it to
    Model.train_on_items; len() is the batch size (Keras' logs['size'])."""
    __slots__ = ('items', 'window')

    def __init__(self, items, window):
        self.items, self.window = np.ascontiguousarray(items, dtype=np.int64), int(window)

    def __len__(self):
        return len(self.items)


# ---- callbacks (keras.callbacks protocol) -----------------------------------------------------------------
class Callback(object):
    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class CSVLogger(Callback):
    def __init__(self, filename, separator=',', append=False):
        super(CSVLogger, self).__init__()
        self.filename, self.sep, self.append = filename, separator, append
        self.keys = None
        self.fp = None

    def on_train_begin(self, logs=None):
        self.fp = open(self.filename, 'a' if self.append else 'w')

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
            self.writer = csv.DictWriter(self.fp, fieldnames=['epoch'] + self.keys, delimiter=self.sep)
            self.writer.writeheader()
        row = {'epoch': epoch}
        row.update((k, logs.get(k, 'NA')) for k in self.keys)
        self.writer.writerow(row)
        self.fp.flush()

    def on_train_end(self, logs=None):
        if self.fp:
            self.fp.close()
            self.fp = None


class ModelCheckpoint(Callback):
    """keras.callbacks.ModelCheckpoint.  background=True (not in Keras; UNet2DSummary.fit uses it): the model is
    snapshotted when the callback runs -- so the file holds exactly the epoch's weights -- but serialised and written by a
    background thread while the next epoch trains; on_train_end waits for the last file."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, mode='auto', period=1, background=False):
        super(ModelCheckpoint, self).__init__()
        self.filepath, self.monitor, self.verbose, self.save_best_only = filepath, monitor, verbose, save_best_only
        self.background = bool(background)
        self.cmp = np.greater if (mode == 'max' or (mode == 'auto' and ('acc' in monitor or monitor.startswith('fmeasure')))) else np.less
        self.best = -np.inf if self.cmp is np.greater else np.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.cmp(cur, self.best):
                return
            self.best = cur
        if self.verbose:
            print('Epoch %05d: saving model to %s' % (epoch, path))
        if self.background and hasattr(self.model, 'wait_for_saves'):
            self.model.save(path, background=True)
        else:
            self.model.save(path)

    def on_train_begin(self, logs=None):
        if self.background and hasattr(self.model, '_writer_process'):
            self.model._writer_process()        # started now: its interpreter start-up overlaps the first epoch, not the second

    def on_train_end(self, logs=None):
        if hasattr(self.model, 'wait_for_saves'):
            self.model.wait_for_saves()


class ReduceLROnPlateau(Callback):
    """Keras 2.0.6 semantics: improvement iff monitor beats best by `epsilon`; after `patience`
    non-improving epochs lr <- max(lr*factor, min_lr); cooldown 0.  Writes logs['lr']."""

    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', epsilon=1e-4,
                 cooldown=0, min_lr=0):
        super(ReduceLROnPlateau, self).__init__()
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.epsilon, self.cooldown, self.min_lr = epsilon, cooldown, min_lr
        maximize = mode == 'max' or (mode == 'auto' and 'acc' in monitor)
        if maximize:
            self.better = lambda a, b: a > b + self.epsilon
            self.best = -np.inf
        else:
            self.better = lambda a, b: a < b - self.epsilon
            self.best = np.inf
        self.wait = 0
        self.cooldown_counter = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = self.model.optimizer.lr
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.better(cur, self.best):
            self.best = cur
            self.wait = 0
        elif self.cooldown_counter <= 0:
            if self.wait >= self.patience:
                old = self.model.optimizer.lr
                if old > self.min_lr + self.min_lr * 1e-4:
                    self.model.optimizer.lr = max(old * self.factor, self.min_lr)
                    if self.verbose:
                        print('Epoch %05d: reducing learning rate to %s.' % (epoch, self.model.optimizer.lr))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
            self.wait += 1


# ---- the model --------------------------------------------------------------------------------------------
class Model(object):
    def __init__(self, window_shape, nb_filters_base=32, conv_kernel_init='he_normal', prop_dropout_base=0.25,
                 upsampling_or_transpose='transpose', device=None, seed=7535):
        self.config = dict(window_shape=tuple(int(v) for v in window_shape), nb_filters_base=int(nb_filters_base),
                           prop_dropout_base=float(prop_dropout_base), upsampling_or_transpose=str(upsampling_or_transpose))
        self.engine = UNetEngine(self.config['window_shape'], nb_filters_base, prop_dropout_base, device=device, seed=seed,
                                 upsampling=(upsampling_or_transpose != 'transpose'),      # unet_2d_summary.py:155,:160
                                 conv_kernel_init=conv_kernel_init)                        # :149,:157,:165
        self.optimizer = None
        self.loss = None
        self.metrics_names = ['loss']
        self.stop_training = False
        self.history = None

    # -- Keras surface ------------------------------------------------------------------------------------
    @property
    def input_shape(self):
        return (None,) + self.config['window_shape']

    output_shape = input_shape

    def count_params(self):
        return self.engine.n_train + self.engine.n_stats

    def summary(self):
        e = self.engine
        print('UNet2DS (HIP/gfx950)  input %r  nb_filters_base %d' % (self.input_shape, e.nfb))
        for l in e.layers:
            print('  %-4s %-5s %4d -> %-4d  params %d' % (l.name, l.kind, l.cin, l.cout,
                                                          int(np.prod(l.kshape)) + l.cout * (5 if l.kind != 'head' else 1)))
        print('Total params: %d (trainable %d)' % (self.count_params(), e.n_train))

    def compile(self, optimizer, loss='binary_crossentropy', metrics=None):
        name = loss if isinstance(loss, str) else getattr(loss, '__name__', str(loss))
        if name not in LOSS_KINDS:      # the reference's `losses` dict, unet_2d_summary.py:372-377
            raise ValueError('loss %r is not one of %s' % (name, sorted(LOSS_KINDS)))
        self.optimizer = optimizer if optimizer is not None else Adam(0.002)
        self.loss = name
        self.engine.loss_kind = LOSS_KINDS[name]
        self.metrics_names = ['loss'] + list(METRIC_NAMES)

    def get_weights(self):
        return self.engine.get_weights()

    def set_weights(self, weights):
        self.engine.set_weights(weights)

    def copy_weights_from(self, other):
        """set_weights(other.get_weights()) (unet_2d_summary.py:69) without the round trip through 134 host arrays when both
        are HIP models of the same architecture on one GPU: two device-to-device copies (31 MB + moving statistics)."""
        a, b = self.engine, getattr(other, 'engine', None)
        if b is None or a.device != b.device or a.pflat.numel() != b.pflat.numel() or a.sflat.numel() != b.sflat.numel() \
                or a.nfb != b.nfb or a.upsampling != b.upsampling:
            return self.set_weights(other.get_weights())
        with torch.cuda.device(a.device):
            b._settle_tail()
            a.pflat.copy_(b.pflat, non_blocking=True)
            a.sflat.copy_(b.sflat, non_blocking=True)
        a._packed_dirty = True
        a._fold_dirty = True

    def predict(self, x, batch_size=32, verbose=0):
        """x: (N,H,W) float32 -> (N,H,W) float32 probabilities (learning phase 0: moving stats, no dropout)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 3 or tuple(x.shape[1:]) != self.config['window_shape']:
            raise ValueError('expected input of shape (N,%d,%d), got %r' % (self.config['window_shape'] + (x.shape,)))
        out = np.empty(x.shape, np.float32)
        for i in range(0, x.shape[0], batch_size):
            xb = torch.from_numpy(x[i:i + batch_size]).to(self.engine.device)
            out[i:i + batch_size] = self.engine.forward_infer_checked(xb).cpu().numpy()[:xb.shape[0]]
        return out

    def train_on_batch(self, x, y, drop_masks=None, presharded=False):
        """One optimizer step; returns [loss, F1, prec, reca, dice, dicesq, posyt, posyp] (Keras order).
        Under torch.distributed (one process per GPU) every rank is handed the SAME global batch and trains on
        its contiguous slice (presharded=True: x, y are already this rank's slice); gradients and metric sums are
        all-reduced over RCCL.  Host -> device: the slice is staged in pinned memory and copied on a separate HIP
        stream into one of two rotating device buffers, so the transfer of step k+1 overlaps the backward of step k."""
        if self.optimizer is None:
            raise RuntimeError('compile() the model first')
        eng = self.engine
        sl = slice(0, len(x)) if presharded else parallel.shard_slice(len(x))
        xs = np.asarray(x[sl], dtype=np.float32)
        ys = np.asarray(y[sl], dtype=np.uint8)
        if xs.ndim != 3 or tuple(xs.shape[1:]) != self.config['window_shape'] or ys.shape != xs.shape:
            raise ValueError('expected x, y of shape (N,%d,%d), got %r and %r' % (self.config['window_shape'] + (xs.shape, ys.shape)))
        xd, yd, slot = self._stage(xs, ys)
        masks = None
        if drop_masks is not None:
            masks = {k: torch.from_numpy(np.ascontiguousarray(v[sl])).to(eng.device) for k, v in drop_masks.items()}
        out = self.train_on_device_batch(xd, yd, masks)
        slot['done'].record(torch.cuda.current_stream(eng.device))      # the step's last reader of xd / yd (Adam) is queued
        return out

    def train_on_items(self, batch):
        """train_on_batch for a DeviceBatch (UNet2DSummary._device_batch_gen): this rank's items are cut out of the resident
        summaries / masks by dc_crop_augment on the step's own stream -- no pixel crosses PCIe, nothing is staged."""
        if self.optimizer is None:
            raise RuntimeError('compile() the model first')
        if batch.window != self.config['window_shape'][0] or self.config['window_shape'][0] != self.config['window_shape'][1]:
            raise ValueError('batch of %d^2 windows for a model of input shape %r' % (batch.window, self.config['window_shape']))
        with torch.cuda.device(self.engine.device):
            xd, yd = self.engine.crop_batch(batch.items)
            return self._train_on_device_batch(xd, yd, None)

    def _stage(self, xs, ys):
        """numpy slice -> (pinned host buffer -> device buffer) of a 2-deep ring, copied on the copy stream."""
        eng = self.engine
        key = tuple(xs.shape)
        st = getattr(self, '_staging', None)
        if st is None or st['key'] != key:
            with torch.cuda.device(eng.device):
                st = dict(key=key, turn=0, copy=torch.cuda.Stream(device=eng.device), slots=[
                    dict(xh=torch.empty(key, dtype=torch.float32).pin_memory(), yh=torch.empty(key, dtype=torch.uint8).pin_memory(),
                         xd=torch.empty(key, dtype=torch.float32, device=eng.device),
                         yd=torch.empty(key, dtype=torch.uint8, device=eng.device),
                         done=torch.cuda.Event(), ready=torch.cuda.Event()) for _ in range(2)])
            self._staging = st
        slot = st['slots'][st['turn']]
        st['turn'] ^= 1
        slot['done'].synchronize()            # the step that last used this slot (two steps ago) has finished with it
        slot['xh'].numpy()[...] = xs
        slot['yh'].numpy()[...] = ys
        with torch.cuda.device(eng.device), torch.cuda.stream(st['copy']):
            slot['xd'].copy_(slot['xh'], non_blocking=True)
            slot['yd'].copy_(slot['yh'], non_blocking=True)
            slot['ready'].record(st['copy'])
        torch.cuda.current_stream(eng.device).wait_event(slot['ready'])
        return slot['xd'], slot['yd'], slot

    def train_on_device_batch(self, xd, yd, masks=None):
        """train_on_batch for a LOCAL shard already resident in HBM (xd float32 (n,H,W), yd uint8 (n,H,W)):
        forward + BCE + backward + (RCCL all-reduce of the flat gradient and of the 8 metric sums) + Adam."""
        if self.optimizer is None:
            raise RuntimeError('compile() the model first')
        # events, pinned copies and collectives below act on the CURRENT device's streams: make it the engine's
        # (Model(device='cuda:1') while device 0 is current)
        with torch.cuda.device(self.engine.device):
            return self._train_on_device_batch(xd, yd, masks)

    def _train_on_device_batch(self, xd, yd, masks):
        eng = self.engine
        eng.forward_train(xd, yd, masks)
        world = parallel.world_size()
        sync = eng._sync_bn()
        # The loss / metric sums are complete once the forward's head kernel has run: their (all-reduced) copy goes to
        # pinned host memory right away and the host only waits for THAT copy -- backward and Adam of this step are still
        # executing when the call returns, so the next step's launches queue up behind them with no idle gap.
        sums = eng._train_bufs(xd.shape[0])['sums']
        grad_scale = 1.0 / world
        dp = parallel.exchange_active()          # world > 1, or the one-rank RCCL rehearsal (DC_DIST_FORCE=1)
        if dp:
            if sync:
                # 'sync' = ONE device's step on the global batch: the dice losses' backward must see the GLOBAL sums
                # (in place), and their per-pixel gradient carries no 1/count, so the summed gradient is already the
                # global one (BCE kinds divide by the LOCAL pixel count -> mean over ranks)
                parallel.all_reduce_sum(sums)
                if eng.loss_kind >= 2:
                    grad_scale = 1.0
            else:
                # 'local' = the mean of G independent shard losses: each rank's backward keeps its own sums
                sums = sums.clone()
                parallel.all_reduce_sum(sums)
        if getattr(self, '_sums_host', None) is None:
            self._sums_host = torch.empty(sums.shape, dtype=sums.dtype).pin_memory()
        self._sums_host.copy_(sums, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        if dp:
            self._backward_allreduce()
        else:
            eng.backward(defer_tail=True)      # adam_step() joins the weight-gradient stream (128^2 x 20 step: 3.19 -> 3.15 ms)
        o = self.optimizer
        eng.adam_step(o.lr, o.beta_1, o.beta_2, o.epsilon, grad_scale=grad_scale)
        copied.synchronize()
        m = metrics_from_sums(self._sums_host.numpy().copy(), float(world * xd.shape[0] * xd.shape[1] * xd.shape[2]), self.loss)
        return [m[k] for k in self.metrics_names]

    def _backward_allreduce(self):
        """Backward + gradient all-reduce (RCCL over xGMI; gloo in the CPU-launched tests).  The flat gradient goes
        out in the three contiguous ranges of UNetEngine.grad_buckets(): decoder + head (39 % of the 31 MB) and the
        bottleneck (46 %) are reduced while the encoder's backward -- the long 256^2 / 512^2 layers -- is still running;
        only the encoder's 15 % is exposed.  DC_AR_BUCKETS=1: one blocking all-reduce after the backward."""
        import torch.distributed as dist
        eng = self.engine
        comm = parallel.native_comm(eng.device)
        if comm is not None:
            # RCCL through the C ABI (dc_comm_all_reduce_sum on the engine's collective stream): the exchange is part of the
            # backward's launch sequence -- one tape, no return to Python between the ranges (DC_COMM=torch: the path below)
            eng.ar_probe = getattr(self, 'ar_events', None)     # bench.py: events around the exposed part (the last range + the join)
            eng.backward(comm=comm)
            return
        if os.environ.get('DC_AR_BUCKETS', '3') == '1':
            eng.backward()
            t0 = self._ar_mark()
            parallel.all_reduce_sum(eng.gflat)
            self._ar_mark(t0)
            return
        works = []
        eng.backward(bucket_cb=lambda lo, hi: works.append(dist.all_reduce(eng.gflat[lo:hi], async_op=True)))
        t0 = self._ar_mark()
        lo, hi = eng.grad_buckets()[-1]
        dist.all_reduce(eng.gflat[lo:hi])
        for w in works:
            w.wait()
        self._ar_mark(t0)

    def _ar_mark(self, start=None):
        """bench.py instrumentation: HIP events around the part of the gradient exchange the step actually waits for."""
        rec = getattr(self, 'ar_events', None)
        if rec is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            rec.append((start, ev))
        return ev

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, max_queue_size=10,
                      initial_epoch=0):
        """Keras 2.0.6 fit_generator: one background generator THREAD feeding a queue of depth
        `max_queue_size`; per-epoch logs are batch-size-weighted means; callbacks' on_epoch_end run in list order
        on the caller's thread sharing ONE mutable logs dict; History last."""
        self.history = History()
        cbs = list(callbacks or []) + [self.history]
        for cb in cbs:
            cb.set_model(self)
            cb.set_params(dict(epochs=epochs, steps=steps_per_epoch, verbose=verbose, metrics=self.metrics_names))
        q = queue.Queue(maxsize=max(1, max_queue_size))
        stop = threading.Event()
        # a generator that already yields this rank's slice of the global batch (UNet2DSummary._batch_gen under DP)
        presharded = bool(getattr(generator, 'presharded', False))

        def producer():
            try:
                while not stop.is_set():
                    item = next(generator)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
            except Exception as e:      # surfaced on the consumer side
                while not stop.is_set():
                    try:
                        q.put(e, timeout=0.1)
                        break
                    except queue.Full:
                        continue

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        self.stop_training = False
        for cb in cbs:
            cb.on_train_begin({})
        try:
            for epoch in range(initial_epoch, epochs):
                for cb in cbs:
                    cb.on_epoch_begin(epoch, {})
                totals = dict((k, 0.0) for k in self.metrics_names)
                seen = 0
                tic = time.time()
                for step in range(steps_per_epoch):
                    item = q.get()
                    if isinstance(item, Exception):
                        raise item
                    xb = item if isinstance(item, DeviceBatch) else item[0]
                    blogs = {'batch': step, 'size': len(xb)}
                    for cb in cbs:
                        cb.on_batch_begin(step, blogs)
                    if isinstance(item, DeviceBatch):           # items for dc_crop_augment: already this rank's slice
                        vals = self.train_on_items(item)
                    else:
                        vals = self.train_on_batch(xb, item[1], presharded=presharded)
                    for k, v in zip(self.metrics_names, vals):
                        blogs[k] = v
                        totals[k] += v * len(xb)
                    seen += len(xb)
                    for cb in cbs:
                        cb.on_batch_end(step, blogs)
                logs = dict((k, totals[k] / max(seen, 1)) for k in self.metrics_names)
                if verbose and parallel.rank() == 0:
                    print('Epoch %d/%d - %.1fs - %s' % (epoch + 1, epochs, time.time() - tic,
                                                        ' - '.join('%s: %.4f' % (k, logs[k]) for k in self.metrics_names)))
                for cb in cbs:
                    cb.on_epoch_end(epoch, logs)
                if self.stop_training:
                    break
        finally:
            stop.set()
            # the generator thread must be gone before the interpreter can exit: a daemon thread still inside native code
            # (numpy, a queue wait) when Python finalises is unwound by force -- 'terminate called without an active
            # exception', SIGABRT after a run that had finished fine (seen once in ~10 runs of the 2-rank example)
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=30)
            for cb in cbs:
                cb.on_train_end({})
        return self.history

    # -- checkpoints ------------------------------------------------------------------------------------------------------
    # '*.hdf5' / '*.h5' (what ModelCheckpoint passes, unet_2d_summary.py:423): the reference's Keras-2.0.x model-file layout,
    # written in-process (keras_io / hdf5_min); anything else: the build's own .npz container.  Loading sniffs the file.
    def _optimizer_state(self, m=None, v=None, iterations=None, config=None):
        eng = self.engine
        m = eng.mflat.cpu().numpy() if m is None else m
        v = eng.vflat.cpu().numpy() if v is None else v
        ms, vs = [], []
        for l in eng.layers:
            for key in ('k', 'b', 'gamma', 'beta'):
                if key in l.off:
                    o, shp = l.off[key]
                    n = int(np.prod(shp))
                    ms.append(m[o:o + n].reshape(shp).copy())
                    vs.append(v[o:o + n].reshape(shp).copy())
        return dict(config=config if config is not None else self.optimizer.get_config(),
                    iterations=int(eng.iterations if iterations is None else iterations), m=ms, v=vs)

    def _set_optimizer_state(self, st, loss):
        eng = self.engine
        m, v = np.zeros(eng.mflat.numel(), np.float32), np.zeros(eng.vflat.numel(), np.float32)
        i = 0
        for l in eng.layers:
            for key in ('k', 'b', 'gamma', 'beta'):
                if key in l.off:
                    o, shp = l.off[key]
                    n = int(np.prod(shp))
                    if tuple(np.shape(st['m'][i])) != tuple(shp):
                        raise ValueError('optimizer state %d: shape %r != %r' % (i, np.shape(st['m'][i]), shp))
                    m[o:o + n] = np.asarray(st['m'][i], np.float32).ravel()
                    v[o:o + n] = np.asarray(st['v'][i], np.float32).ravel()
                    i += 1
        eng.mflat.copy_(torch.from_numpy(m))
        eng.vflat.copy_(torch.from_numpy(v))
        eng.iterations = int(st['iterations'])
        oc = dict((k, st['config'][k]) for k in ('lr', 'beta_1', 'beta_2', 'epsilon') if k in st.get('config', {}))
        self.compile(Adam(**oc), loss or 'binary_crossentropy')

    def save(self, filepath, include_optimizer=True, background=False):
        """background=True: the parameters, moving statistics and Adam state are copied device -> pinned host memory on the
        current stream (a snapshot of THIS moment: ~95 MB, a few ms of copy-engine time); a feeder thread dumps them raw
        (one GIL-free write) and the checkpoint-writer PROCESS (deep_calcium_amd/_ckpt_writer.py) serialises the file while
        training goes on, off this process's GIL.  wait_for_saves() collects the result (one snapshot in flight at a time)."""
        with_opt = include_optimizer and self.optimizer is not None
        self.wait_for_saves()
        snap = self._snapshot(with_opt, sync=not background)
        if not background:
            from ._ckpt_writer import write_checkpoint
            b = snap['bufs']
            write_checkpoint(filepath, self.config, b['p'].numpy(), b['s'].numpy(), b['m'].numpy() if with_opt else None,
                             b['v'].numpy() if with_opt else None, snap['meta'])
            return
        import threading
        box = []

        def feed():
            raw = None
            try:
                snap['event'].synchronize()
                proc = self._writer_process()
                b = snap['bufs']
                parts = [b['p'].numpy(), b['s'].numpy()] + ([b['m'].numpy(), b['v'].numpy()] if with_opt else [])
                raw = os.path.join('/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else
                                   (os.path.dirname(os.path.abspath(filepath)) or '.'),
                                   'dcunet_ckpt_%d_%d.raw' % (os.getpid(), id(self)))
                with open(raw, 'wb') as fp:
                    for a in parts:
                        a.tofile(fp)
                job = dict(raw=raw, path=os.path.abspath(filepath), config=self.config, meta=snap['meta'],
                           sizes=[int(a.size) for a in parts] + ([0, 0] if not with_opt else []))
                proc.stdin.write((json.dumps(job) + '\n').encode())
                proc.stdin.flush()
                reply = proc.stdout.readline().decode().strip()
                if not reply.startswith('ok '):
                    raise IOError(reply or 'the checkpoint writer process died')
            except BaseException as e:       # surfaced by wait_for_saves()
                box.append(e)
                if raw is not None and os.path.exists(raw):      # (the writer process removes it when it got that far)
                    try:
                        os.remove(raw)
                    except OSError:
                        pass
        th = threading.Thread(target=feed, daemon=False)
        self._saving = (th, box, filepath)
        th.start()

    def _writer_process(self):
        proc = getattr(self, '_writer', None)
        if proc is None or proc.poll() is not None:
            import subprocess
            import sys
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
            # a child that only ever runs numpy: it must never open the GPU (HIP_VISIBLE_DEVICES hides it anyway)
            env['HIP_VISIBLE_DEVICES'] = ''
            proc = self._writer = subprocess.Popen([sys.executable, '-m', 'deep_calcium_amd._ckpt_writer'], stdin=subprocess.PIPE,
                                                   stdout=subprocess.PIPE, env=env, close_fds=True)
            import atexit
            import weakref
            ref = weakref.ref(proc)

            def stop():
                pr = ref()
                if pr is not None and pr.poll() is None:
                    try:
                        pr.stdin.close()
                        pr.wait(timeout=10)
                    except Exception:
                        pr.kill()
            atexit.register(stop)
        return proc

    def wait_for_saves(self):
        pending, self._saving = getattr(self, '_saving', None), None
        if pending is not None:
            th, box, path = pending
            th.join()
            if box:
                raise IOError('writing %s failed: %r' % (path, box[0]))

    def _snapshot(self, with_opt, sync=False):
        """Host copy of everything a checkpoint holds, as of THIS point of the engine's stream.  Two hops: device -> device
        scratch on the current stream (124 MB at HBM speed: ~0.1 ms in front of the next step), then scratch -> pinned host
        memory on a separate copy stream (~4 ms of PCIe that the next epoch's first steps no longer wait for).  Both sets of
        buffers are reused from save to save (wait_for_saves() has made sure the previous writer is done with them)."""
        eng = self.engine
        with torch.cuda.device(eng.device):
            eng._settle_tail()
            srcs = (('p', eng.pflat), ('s', eng.sflat)) + ((('m', eng.mflat), ('v', eng.vflat)) if with_opt else ())
            bufs = getattr(self, '_snap_bufs', None)
            if bufs is None:
                bufs = self._snap_bufs = dict((k, torch.empty(t.numel(), dtype=t.dtype).pin_memory()) for k, t in
                                              (('p', eng.pflat), ('s', eng.sflat), ('m', eng.mflat), ('v', eng.vflat)))
                self._snap_dev = dict((k, torch.empty_like(t)) for k, t in
                                      (('p', eng.pflat), ('s', eng.sflat), ('m', eng.mflat), ('v', eng.vflat)))
                self._snap_stream = torch.cuda.Stream(device=eng.device)
            main = torch.cuda.current_stream(eng.device)
            for k, t in srcs:
                self._snap_dev[k].copy_(t, non_blocking=True)
            self._snap_stream.wait_stream(main)
            with torch.cuda.stream(self._snap_stream):
                for k, t in srcs:
                    bufs[k].copy_(self._snap_dev[k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._snap_stream)
            if sync:
                ev.synchronize()
        meta = dict(iterations=int(eng.iterations), loss=self.loss, metrics=list(self.metrics_names[1:]),
                    opt_config=self.optimizer.get_config() if with_opt else None, compiled=self.optimizer is not None)
        return dict(event=ev, bufs=bufs, with_opt=with_opt, meta=meta)

    def load_state(self, filepath, with_optimizer):
        state = read_checkpoint(filepath)
        self.set_weights(state['weights'])
        if with_optimizer and state.get('optimizer') is not None:
            self._set_optimizer_state(state['optimizer'], state.get('loss'))
        return state


def read_checkpoint(path):
    """Either checkpoint container -> dict(weights, config, optimizer | None, loss): a Keras-2.0.x HDF5 model file (the
    reference's format: released weights, ModelCheckpoint files, files written by Model.save('*.hdf5')) or the build's
    own .npz.  The format is sniffed from the file's first bytes, not from its name."""
    from . import hdf5_min, keras_io
    if hdf5_min.is_hdf5(path):
        return keras_io.read_keras_model(path)
    with open(path, 'rb') as fp:
        magic = fp.read(4)
    if magic[:2] != b'PK':
        raise ValueError('%s is neither a Keras HDF5 model file nor a dcunet .npz checkpoint' % path)
    z = np.load(path)
    if 'meta' not in z.files:
        raise ValueError('%s is not a dcunet checkpoint (no meta record)' % path)
    meta = json.loads(bytes(z['meta']).decode())
    if meta.get('format') != 'dcunet-npz-1':
        raise ValueError('%s is not a dcunet checkpoint' % path)
    n = len([k for k in z.files if k.startswith('w_')])
    out = dict(weights=[z['w_%03d' % i] for i in range(n)], config=meta['config'], optimizer=None, loss=meta.get('loss'))
    if 'optimizer' in meta:
        oc = dict(meta['optimizer'])
        it = int(oc.pop('iterations'))
        m, v = z['opt_m'], z['opt_v']
        ms, vs, o = [], [], 0
        from .layers import build_layer_table
        cfg = meta['config']
        for l in build_layer_table(cfg['nb_filters_base'], cfg.get('prop_dropout_base', 0.25),
                                   cfg.get('upsampling_or_transpose', 'transpose') != 'transpose'):
            shapes = [l.kshape, (l.cout,)] + ([(l.cout,), (l.cout,)] if l.kind != 'head' else [])
            for shp in shapes:
                k = int(np.prod(shp))
                ms.append(m[o:o + k].reshape(shp))
                vs.append(v[o:o + k].reshape(shp))
                o += k
        out['optimizer'] = dict(config=oc, iterations=it, m=ms, v=vs)
    return out


def unet_hip(window_shape=(128, 128), nb_filters_base=32, conv_kernel_init='he_normal',
             prop_dropout_base=0.25, upsampling_or_transpose='transpose'):
    """Drop-in for the reference's net_builder_func `unet()` (unet_2d_summary.py:123-124): same signature,
    returns an uncompiled model whose ops are HIP kernels."""
    return Model(window_shape, nb_filters_base, conv_kernel_init, prop_dropout_base, upsampling_or_transpose)


def load_model_with_new_input_shape(model_path, input_shape, **load_model_args):
    """The weight-I/O seam of /root/reference/deepcalcium/utils/keras_helpers.py:24-68: the same weights at a
    new window size (the net is fully convolutional).  `compile=True` restores the optimizer state.  model_path may be
    a Keras-2.0.x HDF5 model file (the reference's own format, e.g. the released unet2ds_model.hdf5 or a
    ModelCheckpoint file -- read in-process, optimizer_weights included) or the build's .npz."""
    state = read_checkpoint(model_path)
    cfg = state['config']
    model = Model(tuple(input_shape), cfg['nb_filters_base'], conv_kernel_init=None,      # weights come from the file
                  prop_dropout_base=cfg['prop_dropout_base'],
                  upsampling_or_transpose=cfg.get('upsampling_or_transpose', 'transpose'))
    model.set_weights(state['weights'])
    if bool(load_model_args.get('compile', True)) and state.get('optimizer') is not None:
        model._set_optimizer_state(state['optimizer'], state.get('loss'))
    return model
