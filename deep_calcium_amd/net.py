"""UNet2DS execution engine: orchestrates the libdcunet HIP kernels for one GPU.

Mirrors the graph built by unet() at
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:123-224 (topology, layer order, dropout
placement, skip-concat order) -- the arithmetic itself is entirely in include/dcunet.h entry points.
torch is used only as the device-memory container (torch.empty / data_ptr) and for the current stream.

Memory plan (sized for 288 GB HBM: nothing is recomputed, nothing is pooled):
  * all trainable parameters live in ONE flat fp32 buffer `pflat` in Keras get_weights() order
    (kernel, bias, gamma, beta per layer) with matching flat `gflat` / Adam `mflat`,`vflat`, so the
    optimizer is one launch and data-parallel training all-reduces one 31 MB message;
  * skip-concat is free: the encoder block and the up-conv block of a level write straight into the two
    channel halves of one `cat` buffer through the kernels' pixel-stride (ld) arguments;
  * training keeps z (pre-BN) and the block outputs for backward; inference folds BN into the conv epilogue.
"""
import collections
import os
import threading

import numpy as np
import torch

from . import parallel
from ._lib import lib, DcunetError, Tape, Var, ops_equal

BN_EPS = 1e-3


def _ptr(t, offset_floats=0):
    return t.data_ptr() + 4 * offset_floats


def _on_device(fn):
    """The C ABI launches on the CURRENT HIP device (dcunet.h: 'caller sets the device'): make the engine's device
    current for the duration of the call, so Model(device='cuda:1') works while device 0 is current."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **k)
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


from .layers import LayerSpec, build_layer_table, assign_offsets, split_weights      # noqa: E402,F401  (torch-free: the checkpoint writer process imports them)


_PINNED = collections.OrderedDict()      # key -> [pinned tensor, event of the last async H2D copy that read it | None]
_PINNED_MAX_BYTES = 512 << 20
_PINNED_LOCK = threading.Lock()          # concurrent predict() calls (one key set per thread) share the LRU bookkeeping


def _pinned(tag, shape, dtype, entry=False):
    """Page-locked host buffers outlive engines: predict() builds a new engine per call (it loads a model file, as the
    reference does) and pinning costs ~9 ms per buffer.  Least-recently-used entries are dropped beyond 512 MB (every new
    dataset count / window size / calling thread adds a key); torch's pinned allocator defers the actual release past any
    copy still in flight.  entry=True returns the [tensor, last-H2D-event] cell itself, so the "has the previous copy out of
    this buffer finished" guard survives the job (and the exception) that issued the copy."""
    key = (tag, tuple(shape), dtype, threading.get_ident())      # one set per calling thread: no sharing between concurrent predict()s
    with _PINNED_LOCK:
        ent = _PINNED.get(key)
        if ent is not None:
            _PINNED.move_to_end(key)
            return ent if entry else ent[0]
    new = [torch.empty(tuple(shape), dtype=dtype).pin_memory(), None]       # ~9 ms: outside the lock
    with _PINNED_LOCK:
        ent = _PINNED.setdefault(key, new)
        total = sum(e[0].numel() * e[0].element_size() for e in _PINNED.values())
        while total > _PINNED_MAX_BYTES and len(_PINNED) > 1:
            k0 = next(iter(_PINNED))
            if k0 == key:
                break
            old = _PINNED.pop(k0)
            total -= old[0].numel() * old[0].element_size()
    return ent if entry else ent[0]


class _TtaJob(object):
    """predict() over several datasets WITHOUT a host synchronisation per dataset: enqueue(i, ...) stages image i in one
    of two pinned slots, launches gather -> one batch-K forward -> merge/crop/threshold -> D2H of the mask and returns;
    finish() synchronises once and hands back the masks.  The host reads / pads dataset i+1 while the device works on i.
    The optimistic fp16 range flag of every forward is kept per dataset; flagged ones (never on trained weights) are
    redone in measured mode by finish().  augmentations None = the plain forward (K = 1, identity map)."""

    def __init__(self, eng, count, augmentations):
        self.eng, self.count = eng, int(count)
        self.aug = list(augmentations) if augmentations is not None else [('identity', lambda a: a, lambda a: a)]
        K, H, W = len(self.aug), eng.H, eng.W
        n = H * W
        dev = eng.device
        key = ('ttamaps', K, tuple(name for name, _, _ in self.aug))
        m = eng._bufs.get(key)
        if m is None:
            idx = np.arange(n, dtype=np.int32).reshape(1, H, W)
            fwd = np.stack([np.ascontiguousarray(f(idx)).reshape(-1) for _, f, _ in self.aug])
            inv = np.stack([np.ascontiguousarray(g(idx)).reshape(-1) for _, _, g in self.aug])
            if fwd.shape != (K, n) or inv.shape != (K, n):
                raise ValueError('test-time augmentations must map (N,H,W) to (N,H,W) on a square window')
            m = eng._bufs[key] = dict(fwd=torch.from_numpy(fwd).to(dev), inv=torch.from_numpy(inv).to(dev),
                                      x=torch.empty((K, H, W), dtype=torch.float32, device=dev),
                                      src=[torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)])
        self.m = m
        self.host = [_pinned('tta_img%d' % k, (H, W), torch.float32, entry=True) for k in range(2)]
        self.mask = torch.empty((self.count, n), dtype=torch.uint8, device=dev)
        self.mask_host = _pinned('tta_mask', (max(self.count, 1), n), torch.uint8)
        self.ovf = torch.zeros(max(self.count, 1), dtype=torch.float32, device=dev)
        self.jobs = [None] * self.count

    def enqueue(self, i, img, hs, ws, threshold):
        eng, m = self.eng, self.m
        K, H, W = len(self.aug), eng.H, eng.W
        with torch.cuda.device(eng.device):
            L, st = eng.L, eng._stream()
            k = i & 1
            slot = self.host[k]
            if slot[1] is not None:
                slot[1].synchronize()                    # the H2D copy that last read this pinned slot (this job or an earlier one) has finished
            slot[0].numpy()[...] = img
            m['src'][k].copy_(slot[0].view(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            slot[1] = ev
            L.dc_gather_maps(_ptr(m['src'][k]), m['fwd'].data_ptr(), _ptr(m['x']), K, H * W, st)
            p = eng.forward_infer(m['x'])
            if eng.mfma == 'f16x3' and eng.range_guard and not eng.infer_measured:
                self.ovf[i:i + 1].copy_(eng._ovf[0:1])
            L.dc_tta_merge(_ptr(p), m['inv'].data_ptr(), K, H, W, int(hs), int(ws), float(threshold),
                           self.mask[i].data_ptr(), None, st)
            self.mask_host[i, :hs * ws].copy_(self.mask[i, :hs * ws], non_blocking=True)
        self.jobs[i] = (np.array(img, dtype=np.float32, copy=True), int(hs), int(ws), float(threshold))

    def finish(self):
        eng = self.eng
        torch.cuda.current_stream(eng.device).synchronize()
        redo = [i for i, f in enumerate(self.ovf.cpu().numpy()[:self.count]) if f != 0.0 and self.jobs[i] is not None]
        out = []
        for i, j in enumerate(self.jobs):
            if j is None:
                out.append(None)
                continue
            _, hs, ws, _ = j
            out.append(self.mask_host[i, :hs * ws].numpy().reshape(hs, ws).copy())
        if redo:                                         # an activation left fp16's range: measured bounds, one by one
            eng.infer_measured = True
            for i in redo:
                img, hs, ws, thr = self.jobs[i]
                out[i] = eng.predict_tta(img, self.aug, hs, ws, thr)
        return out


class UNetEngine(object):
    # Rotation depth of the backward's per-block buffers (dz / dz-on-load table / activation gradient).  The weight gradients run on
    # a side stream, so a buffer may only be rewritten once the weight gradient that read it has finished: with 3 buffers the main
    # stream WAITS on that weight gradient's event three blocks later -- always long satisfied, and still a cross-queue barrier
    # packet that costs the main queue ~12 us each time (`scripts/backward_chain.py`: 33 such gaps per step, 0.28 ms at 512^2 x 16,
    # 0.36 ms of a 3.1-ms step at 128^2 x 20).  With one buffer per block (SLOTS_DEEP >= the ~26 blocks / hand-overs of a step) no
    # buffer is reused inside a step and the waits disappear -- MEASURED (round 5, same-box A/B, scripts/ab_engine_flags.py
    # deep_slots): 3.094 -> 3.105 ms at 128^2 x 20, 17.996 -> 18.108 ms at 512^2 x 16, 3.282 -> 3.245 ms at 96^2 x 32: the gaps are
    # not the waits (the event RECORD on the main queue in front of them stays), and 34 GB more footprint costs more than it saves.
    # Off by default (`deep_slots`); falls back to 3 anyway when it would exceed SLOTS_DEEP_MAX_BYTES.
    SLOTS = 3
    # BatchNorm partial rows: launches that leave >= FOLD_MIN rows (one per pixel tile: the 512^2-class layers outside the role-split
    # kernel, the up-convolutions) fold them to FOLD_ROWS with coalesced whole-row reads first; the finalize's own reads are 16-byte
    # pieces one row apart (77 us for the 16 384 x 4 rows of u0 at 512^2 x 16, on the critical path).  Same box: 17.77 -> 17.67 ms at
    # 512^2 x 16 (0 = never: A/B); the 128^2 / 96^2 steps stay below the threshold (an extra launch costs them 0.2 %).
    FOLD_MIN = int(os.environ.get('DC_STATS_FOLD_MIN', '8192'))
    FOLD_ROWS = 128
    SLOTS_DEEP = 32
    SLOTS_DEEP_MAX_BYTES = 48 << 30

    def __init__(self, window_shape, nb_filters_base=32, prop_dropout_base=0.25, device=None, seed=7535, mfma=None,
                 upsampling=False, bn_mode=None, conv_kernel_init='he_normal'):
        # mfma: 'f16x3' (default; fp32-grade split-fp16 products on the fp16 matrix cores) or 'f32' (fp32 MFMA)
        self.mfma = mfma or os.environ.get('DC_MFMA', 'f16x3')
        if self.mfma not in ('f16x3', 'f32'):
            raise ValueError("mfma must be 'f16x3' or 'f32', got %r" % self.mfma)
        self.streams = 1 if os.environ.get('DC_STREAMS', '2') == '1' else 2     # 2: weight gradients on a side stream
        # BN + ReLU on load: activations that only feed a conv / conv-transpose / the head (no dropout, pool or
        # skip) are never written; their consumers take (z, scale, shift) instead.  f16x3 kernels only.
        self.bnin = self.mfma == 'f16x3' and os.environ.get('DC_BNIN', '1') == '1'
        # dz on load (round 4): the BatchNorm-backward apply pass of a Dropout-free conv block is never run -- its data- and
        # weight-gradient kernels form dz from (da, z) and a per-channel table while they stage their operands
        # (dc_bn_bwd_finalize_dzin / dc_conv3x3_dgrad_dzin_f16x3 / dc_conv3x3_wgrad_dzin_f16x3).  f16x3 kernels only.
        # DC_DZIN = 1 (default): the 512^2-class blocks (level 0), where every kernel of the block is HBM-bound and the pass
        # removed is pure time; 'all': every eligible block (levels >= 1 measured 7-15 % slower matrix kernels for a 30-150 us
        # apply pass that the weight-gradient stream hides anyway: DESIGN 5e); 0: off.
        _dz = os.environ.get('DC_DZIN', '1')
        self.dzin = self.mfma == 'f16x3' and _dz != '0'
        self.dzin_lvls = frozenset(range(5)) if _dz == 'all' else frozenset((0,))
        self.dzin_all = _dz == 'all'
        # the joint data- + weight-gradient kernel for the 32 -> 32 blocks among them (DC_DZIN=2: dz on load, separate kernels)
        self.joint = self.dzin and _dz != '2'
        self._head_bwd_done = False
        # fp16 range guard of the activation operands of the f16x3 contractions (csrc/common.h)
        self.range_guard = self.mfma == 'f16x3'
        # Inference starts OPTIMISTIC: no activation scale (a BatchNorm network's activations are O(1)), the kernels only
        # flag an output beyond fp16's range; forward_infer_checked() then repeats the pass with MEASURED per-channel
        # bounds and keeps doing so for this engine (un-normalised inputs, exotic weights).  DC_INFER_GUARD=1: measured
        # bounds from the first call.
        self.infer_measured = os.environ.get('DC_INFER_GUARD', '0') == '1'
        # BatchNorm under batch-sharded data parallelism: 'local' = each rank normalises over its own shard
        # (standard DP semantics); 'sync' = the per-channel sums are all-reduced in forward and backward, so G ranks x B
        # images reproduce ONE device's step on the G*B batch
        bn_mode = bn_mode or os.environ.get('DC_BN_MODE', 'local')
        if bn_mode not in ('local', 'sync'):
            raise ValueError("bn_mode must be 'local' or 'sync', got %r" % (bn_mode,))
        self.bn_mode = bn_mode
        self.loss_kind = 0      # 0 binary_crossentropy, 1 weighted_binary_crossentropy, 2 dice_loss, 3 dicesq_loss
        H, W = window_shape
        if H % 16 or W % 16:
            raise ValueError('window_shape must be a multiple of 16 (4 max-pools), got %r' % (window_shape,))
        nfb = nb_filters_base
        if nfb < 4 or nfb & (nfb - 1) or nfb > 64:
            raise ValueError('nb_filters_base must be a power of two in [4, 64]')
        if not torch.cuda.is_available():
            raise DcunetError('no GPU visible: the UNet2DS engine has no CPU fallback')
        self.L = lib()
        self.H, self.W, self.nfb, self.drp = H, W, nfb, float(prop_dropout_base)
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise DcunetError('the UNet2DS engine runs on a GPU only, got device %r' % (device,))
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.upsampling = bool(upsampling)
        self.layers = build_layer_table(nfb, self.drp, self.upsampling)
        # dropout after the up layer of each level (unet_2d_summary.py:198,204,210,216)
        self.up_drop = {3: 2 * self.drp, 2: 2 * self.drp, 1: 2 * self.drp, 0: self.drp}
        self.by_name = {l.name: l for l in self.layers}
        nm = []
        if self.bnin:
            nm += ['e%da' % i for i in range(4)] + ['ba'] + ['d%da' % i for i in range(4)] + ['d0b']
            if not self.upsampling:
                nm += ['bb', 'd3b', 'd2b', 'd1b']          # feed a Conv2DTranspose
        self.nm = frozenset(n for n in nm if self.by_name[n].drop <= 0.0)
        # ---- flat parameter layout -----------------------------------------------------------------
        off, soff = assign_offsets(self.layers)
        self.n_train = off
        self.n_stats = soff
        npad = (off + 3) // 4 * 4
        dev = self.device
        self.pflat = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.mflat = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.vflat = torch.zeros(npad, dtype=torch.float32, device=dev)
        self.sflat = torch.zeros(soff, dtype=torch.float32, device=dev)
        # per-layer batch statistics / folded affine (scale, shift), each [cout]
        self.bstat = torch.zeros(6 * soff // 2 + 8, dtype=torch.float32, device=dev)
        self._stat_off = {}
        o = 0
        for l in self.layers:
            if l.kind != 'head':
                self._stat_off[l.name] = o   # mean, invstd, folded scale, shift, training scale, shift : 6 * cout
                o += 6 * l.cout
        # packed weights
        self.wp_fwd, self.wp_dgrad = {}, {}
        for l in self.layers:
            if l.kind == 'head' or (l.kind == 'conv' and l.cin == 1):
                continue
            n = int(np.prod(l.kshape))
            if self.mfma == 'f16x3':
                # pre-split fp16 (hi, lo) slabs: [tap][K/8][hi|lo][col][8], K padded to a multiple of 8
                sz = self.L.dc_pack_weights_f16x3_floats
                fw, bw = ((9, l.cin, l.cout), (9, l.cout, l.cin)) if l.kind == 'conv' else ((1, l.cin, 4 * l.cout), (4, l.cout, l.cin))
                self.wp_fwd[l.name] = torch.empty(sz(*fw), dtype=torch.float32, device=dev)
                self.wp_dgrad[l.name] = torch.empty(sz(*bw), dtype=torch.float32, device=dev)
            else:
                self.wp_fwd[l.name] = torch.empty(n, dtype=torch.float32, device=dev)
                self.wp_dgrad[l.name] = torch.empty(n, dtype=torch.float32, device=dev)
        # fp16 range guard of the f16x3 contractions (csrc/common.h dc_block_guard_scale): one per-channel magnitude
        # bound per activation tensor, written by the layer that produces it (training: |gamma|*sqrt(count)+|beta| from
        # the BatchNorm finalize / apply kernels; inference: the measured max |a| from the conv epilogues) and read by
        # the consumers, in THEIR input-channel order: the two producers of a skip-concat buffer share one array.
        # Every array has DC_ABOUND_SLOTS = 8 replicas `ld` floats apart: training writes / reads replica 0 only, the
        # inference epilogues spread their atomic max over the replicas (one hot address per channel otherwise).
        self._ab_off, self._ab_ld = {}, {}
        o = 0
        for lvl in range(4):
            self._ab_off['cat%d' % lvl], self._ab_ld['cat%d' % lvl] = o, self._cup(lvl) + (nfb << lvl)
            o += 8 * self._ab_ld['cat%d' % lvl]
        for l in self.layers:
            if l.kind != 'head':
                self._ab_off[l.name], self._ab_ld[l.name] = o, l.cout
                o += 8 * l.cout
        self.abound = torch.zeros(o, dtype=torch.float32, device=dev)
        self._ovf = torch.zeros(4, dtype=torch.float32, device=dev)
        self._bufs = {}
        # launch tapes (csrc/tape.cpp): the enqueue sequence of each phase of a steady-state train step, recorded once (and
        # verified against a second recording) per key, then replayed from C.  DC_TAPES=0: every launch from Python.
        # levels >= 1: dz formed by the data gradient AND written once for a plain weight gradient (no apply pass): built, parity-green,
        # measured SLOWER same box (17.651 -> 18.103 ms: the role-split kernel's producers become its longest role, as for DC_DZIN=all)
        self.dz_writeback = False
        self.deep_slots = False           # A/B (set before the first step): one backward buffer set per block instead of a rotation of 3
        self.tail_main = os.environ.get('DC_TAIL_MAIN', '1') == '1'      # the step's last weight gradient on the main stream (A/B: 0)
        self.stats_per_wg = True          # BatchNorm partials: one row per (workgroup, consumer set) of the role-split kernel (A/B: False)
        self.use_tapes = os.environ.get('DC_TAPES', '1') != '0'
        self.tape_verify = int(os.environ.get('DC_TAPE_VERIFY', '0'))     # debug: every K-th replay re-records and compares
        self.ar_buckets = 1 if os.environ.get('DC_AR_BUCKETS', '3') == '1' else 3      # gradient exchange: one all-reduce or three ranges
        self._tapes = {}
        self._tape_epoch = 0
        self.ar_probe = None
        self.tape_replays = 0
        # The weight-gradient stream exists (and has been used once) from construction: HIP multiplexes its streams onto
        # GPU_MAX_HW_QUEUES hardware queues in first-use order, and a side stream first used after the collective library has made its
        # own streams can land on the MAIN stream's queue -- the two then serialise (measured: +0.7...1.0 ms per step; scripts/queue_timeline.py
        # shows every launch on one queue).  deep_calcium_amd/__init__.py also raises GPU_MAX_HW_QUEUES to 8 when the process has not set it.
        self._side_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.Event().record(self._side_stream)
        self._evpool, self._ev_i = [], 0
        self._evpool_f, self._evf_i = [], 0       # events with the system-scope fence: collective boundaries
        self._tail = None
        self._packed_dirty = True
        self._fold_dirty = True
        self.iterations = 0
        self.drop_seed = 0x5eed0000 + seed
        if conv_kernel_init is not None:         # None: the caller sets the weights (checkpoint load): skip the RNG work
            self.set_weights(self.initial_weights(seed, conv_kernel_init))

    # ---- parameters ---------------------------------------------------------------------------------
    def initial_weights(self, seed=7535, conv_kernel_init='he_normal'):
        """Keras-2.0.6 initialisers: `conv_kernel_init` (the reference's `cki`, default he_normal) for
        the 3x3 convolutions and the conv-transposes (unet_2d_summary.py:157,:165), glorot_uniform (Keras' Conv2D
        default) for the head (:221), zero biases, identity BatchNorm.  VarianceScaling family: fans from the kernel
        shape (kh,kw,in,out) -- for Conv2DTranspose's (kh,kw,out,in) kernel Keras therefore takes fan_in = 4*out;
        'normal' = truncated at 2 sigma."""
        rs = np.random.RandomState(seed)

        def tn(shape, std):
            out = rs.standard_normal(shape)
            bad = np.abs(out) > 2
            while bad.any():
                out[bad] = rs.standard_normal(int(bad.sum()))
                bad = np.abs(out) > 2
            return (out * std).astype(np.float32)

        vs = {'he_normal': (2.0, 'fan_in', 'normal'), 'he_uniform': (2.0, 'fan_in', 'uniform'),
              'glorot_normal': (1.0, 'fan_avg', 'normal'), 'glorot_uniform': (1.0, 'fan_avg', 'uniform'),
              'lecun_normal': (1.0, 'fan_in', 'normal'), 'lecun_uniform': (1.0, 'fan_in', 'uniform')}

        def init(name, shape):
            if callable(name):
                return np.asarray(name(shape), np.float32).reshape(shape)
            rf = shape[0] * shape[1]
            fan_in, fan_out = shape[2] * rf, shape[3] * rf
            if name in vs:
                scale, mode, dist = vs[name]
                scale /= max(1.0, {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.0}[mode])
                if dist == 'normal':
                    return tn(shape, np.sqrt(scale))
                lim = np.sqrt(3.0 * scale)
                return rs.uniform(-lim, lim, shape).astype(np.float32)
            if name in ('random_normal', 'normal'):
                return (rs.standard_normal(shape) * 0.05).astype(np.float32)
            if name == 'truncated_normal':
                return tn(shape, 0.05)
            if name in ('random_uniform', 'uniform'):
                return rs.uniform(-0.05, 0.05, shape).astype(np.float32)
            if name in ('zeros', 'zero'):
                return np.zeros(shape, np.float32)
            if name in ('ones', 'one'):
                return np.ones(shape, np.float32)
            raise ValueError('conv_kernel_init %r: not a Keras-2.0.6 initializer this build restates (%s, random_normal, '
                             'random_uniform, truncated_normal, zeros, ones, or a callable shape -> array)'
                             % (name, ', '.join(sorted(vs))))

        W = []
        for l in self.layers:
            W.append(init('glorot_uniform' if l.kind == 'head' else conv_kernel_init, l.kshape))
            W.append(np.zeros(l.cout, np.float32))
            if l.kind != 'head':
                W += [np.ones(l.cout, np.float32), np.zeros(l.cout, np.float32),
                      np.zeros(l.cout, np.float32), np.ones(l.cout, np.float32)]
        return W

    def weight_shapes(self):
        out = []
        for l in self.layers:
            out += [l.kshape, (l.cout,)]
            if l.kind != 'head':
                out += [(l.cout,)] * 4
        return out

    def get_weights(self, p=None, s=None):
        """Keras get_weights(): 134 arrays, [kernel, bias, gamma, beta, moving_mean, moving_variance] per layer.
        p / s: host copies of pflat / sflat taken earlier (a checkpoint snapshot) instead of the live device buffers."""
        p = self.pflat.cpu().numpy() if p is None else p
        s = self.sflat.cpu().numpy() if s is None else s
        return split_weights(self.layers, p, s)

    def set_weights(self, weights):
        shapes = self.weight_shapes()
        if len(weights) != len(shapes):
            raise ValueError('expected %d weight arrays, got %d' % (len(shapes), len(weights)))
        p = np.zeros(self.pflat.numel(), np.float32)
        s = np.zeros(self.n_stats, np.float32)
        i = 0
        for l in self.layers:
            for key in ('k', 'b', 'gamma', 'beta'):
                if key in l.off:
                    o, shp = l.off[key]
                    w = np.asarray(weights[i], np.float32)
                    if tuple(w.shape) != tuple(shp):
                        raise ValueError('weight %d (%s.%s): shape %r != %r' % (i, l.name, key, w.shape, shp))
                    p[o:o + w.size] = w.ravel()
                    i += 1
            if l.kind != 'head':
                for key in ('mmean', 'mvar'):
                    w = np.asarray(weights[i], np.float32)
                    s[l.soff[key]:l.soff[key] + l.cout] = w.ravel()
                    i += 1
        self.pflat.copy_(torch.from_numpy(p))
        self.sflat.copy_(torch.from_numpy(s))
        self._packed_dirty = True
        self._fold_dirty = True

    def pview(self, flat, l, key):
        o, shp = l.off[key]
        return _ptr(flat, o)

    def sview(self, l, key):
        return _ptr(self.sflat, l.soff[key])

    def stat_ptr(self, l, which):
        """which: 0 mean, 1 invstd, 2 scale, 3 shift (inference fold), 4 scale, 5 shift (training batch statistics)"""
        return _ptr(self.bstat, self._stat_off[l.name] + which * l.cout)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # ---- stream choreography through the C ABI (so that a tape records it like any launch) -----------------------------
    def _ev(self):
        """The next ordering-only event of this step: a pool of persistent library events handed out in call order (the k-th
        hand-off of every step uses the same event: a recorded tape stays valid)."""
        i = self._ev_i
        self._ev_i = i + 1
        if i == len(self._evpool):
            import ctypes
            h = ctypes.c_void_p()
            rc = self.L.cdll.dc_event_create_sync(ctypes.byref(h))
            if rc != 0:
                raise DcunetError('dc_event_create_sync failed (%d): %s' % (rc, self.L.cdll.dc_last_error().decode()))
            self._evpool.append(h.value)
        return self._evpool[i]

    def _ev_fenced(self):
        """The same for the hand-offs in front of / behind a COLLECTIVE: events that keep the marker's system-scope fence
        (dc_event_create_fenced) -- the buffer's next reader, or last writer, is a peer GPU over xGMI."""
        i = self._evf_i
        self._evf_i = i + 1
        if i == len(self._evpool_f):
            import ctypes
            h = ctypes.c_void_p()
            rc = self.L.cdll.dc_event_create_fenced(ctypes.byref(h))
            if rc != 0:
                raise DcunetError('dc_event_create_fenced failed (%d): %s' % (rc, self.L.cdll.dc_last_error().decode()))
            self._evpool_f.append(h.value)
        return self._evpool_f[i]

    def _record(self, stream_h, fenced=False):
        ev = self._ev_fenced() if fenced else self._ev()
        self.L.dc_event_record(ev, stream_h)
        return ev

    def _wait(self, stream_h, ev):
        self.L.dc_stream_wait_event(stream_h, ev)

    def _wait_stream(self, dst_h, src_h, fenced=False):
        """work queued on dst from here on runs after everything queued on src so far"""
        self._wait(dst_h, self._record(src_h, fenced))

    def _sync_bn(self):
        """'sync' BatchNorm is live when there is more than one rank -- or under DC_DIST_FORCE=1 (a one-rank process group: the
        44 per-layer all-reduces then run over RCCL inside the real step, which is how their cost is priced on one GPU)."""
        return self.bn_mode == 'sync' and (parallel.world_size() > 1 or parallel.exchange_active())

    def _sync_all_reduce(self, t):
        """One of the per-layer 'sync' BatchNorm messages (<= 8 KB); bench.py brackets them with events (sync_events)."""
        rec = getattr(self, 'sync_events', None)
        if rec is None:
            return parallel.all_reduce_sum(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        parallel.all_reduce_sum(t)
        e1.record()
        rec.append((e0, e1))
        return t

    # Engine state a recorded body branches on WITHOUT it being part of the tape keys: assigning a different value to any of
    # these drops every tape (they bake in the control flow AND raw device pointers), and so does any allocation path that
    # can move a buffer a tape points into (buf() regrow, set_crop_sources, a new side stream).
    _TAPE_STATE = frozenset(('range_guard', 'bnin', 'dzin', 'dzin_lvls', 'dzin_all', 'joint', 'stats_per_wg', 'tail_main',
                             'dz_writeback', 'deep_slots', 'FOLD_MIN', 'bn_mode', 'streams', 'mfma', 'upsampling', 'up_drop',
                             'infer_measured', 'SLOTS', 'SLOTS_DEEP', 'SLOTS_DEEP_MAX_BYTES', '_side_stream', 'nm', 'use_tapes',
                             'ar_buckets'))

    def __setattr__(self, name, value):
        if name in UNetEngine._TAPE_STATE and '_tapes' in self.__dict__:
            old = self.__dict__.get(name, getattr(type(self), name, None))
            if old is not value and old != value:
                self.invalidate_tapes()
        object.__setattr__(self, name, value)

    def invalidate_tapes(self):
        """Forget every recorded launch tape: the next two steps re-record (and re-verify) them."""
        self._tapes.clear()
        self._tape_epoch = self.__dict__.get('_tape_epoch', 0) + 1

    def _v(self, key, value):
        """A per-step launch argument: wrapped for the tape while one is being recorded."""
        return Var(key, value) if self.L.recording() else value

    def _taped(self, key, body, values, post_attrs=(), on_mark=None):
        """Run one phase of a step: body() enqueues it launch by launch (and is what gets recorded); once two consecutive
        recordings under `key` agree, the phase is replayed from C with `values()` = {Var key: this step's value}.  `key` must
        hold every piece of state body()'s control flow reads; post_attrs: attributes body() leaves behind (restored after a
        replay)."""
        if not self.use_tapes or not hasattr(self.L, 'record_begin'):
            return body()
        ent = self._tapes.get(key)
        if ent is not None and ent['tape'] is not None:
            self.tape_replays += 1
            if self.tape_verify and self.tape_replays % self.tape_verify == 0:
                # debug hook (DC_TAPE_VERIFY=K): every K-th replay is a fresh recording instead -- the launches go out one by one --
                # and must equal what the tape holds; a difference means engine state changed without dropping the tapes
                self.L.record_begin()
                try:
                    ret = body()
                finally:
                    ops = self.L.record_end()
                if not ops_equal(ent['ops'], ops):
                    raise DcunetError('launch tape %r no longer matches the sequence the engine issues (DC_TAPE_VERIFY): some state '
                                      'the recorded phase depends on changed without invalidate_tapes()' % (key[0],))
                return ret
            ent['tape'].replay(values(), on_mark)
            for a, v in ent['post'].items():
                setattr(self, a, v)
            return ent['ret']
        self.L.record_begin()
        try:
            ret = body()
        finally:
            ops = self.L.record_end()
        post = dict((a, getattr(self, a)) for a in post_attrs)
        if ent is not None and ent['post'] == post and ops_equal(ent['ops'], ops):
            ent.update(tape=Tape(self.L, ops), ret=ret)
        else:
            self._tapes[key] = dict(ops=ops, tape=None, post=post, ret=ret)
        return ret

    def buf(self, name, numel, dtype=torch.float32):
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            if t is not None:
                self.invalidate_tapes()          # a regrown buffer moves: recorded pointers into the old one are dead
            t = torch.empty(int(numel), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    @_on_device
    def repack(self):
        """HWIO / (2,2,Cout,Cin) kernels -> the [tap][K/4][N][4] layouts of the implicit-GEMM kernels."""
        if not self._packed_dirty:
            return
        L, st = self.L, self._stream()
        jobs = []
        for l in self.layers:
            if l.name not in self.wp_fwd:
                continue
            src = self.pview(self.pflat, l, 'k')
            ci, co = l.cin, l.cout
            if l.kind == 'conv':
                jobs.append((src, _ptr(self.wp_fwd[l.name]), 9, ci, co, ci * co, co, 1, 0))
                jobs.append((src, _ptr(self.wp_dgrad[l.name]), 9, co, ci, ci * co, 1, co, 1))
            else:
                jobs.append((src, _ptr(self.wp_fwd[l.name]), 1, ci, 4 * co, 0, 1, ci, 0))
                jobs.append((src, _ptr(self.wp_dgrad[l.name]), 4, co, ci, co * ci, ci, 1, 0))
        if self.mfma == 'f16x3':
            if getattr(self, '_pack_jobs', None) is None:      # pointers and shapes are fixed: build the table once
                rows, b = [], 0
                for j in jobs:
                    total = j[2] * ((j[3] + 7) // 8) * 8 * j[4]
                    rows.append(list(j) + [b])
                    b += max(1, min((total // 8 + 1023) // 1024, 2048))   # a thread packs 4 of the image's 16-byte slots
                rows.append([0] * 9 + [b])
                self._pack_jobs = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(jobs), b)
            tab, n, blocks = self._pack_jobs
            L.dc_pack_weights_f16x3_batch(tab.data_ptr(), n, blocks, st)
        else:
            for j in jobs:
                L.dc_pack_weights(*(j + (st,)))
        self._packed_dirty = False

    def refold(self):
        if not self._fold_dirty:
            return
        L, st = self.L, self._stream()
        for l in self.layers:
            if l.kind == 'head':
                continue
            L.dc_bn_fold(self.pview(self.pflat, l, 'gamma'), self.pview(self.pflat, l, 'beta'),
                         self.sview(l, 'mmean'), self.sview(l, 'mvar'), self.pview(self.pflat, l, 'b'),
                         BN_EPS, self.stat_ptr(l, 2), self.stat_ptr(l, 3), l.cout, st)
        self._fold_dirty = False

    def _ab_infer(self, l):
        """(in_abound, in_abound_ld, out_absmax, out_absmax_ld) of layer l for the current inference mode."""
        if self.mfma != 'f16x3' or not self.range_guard:
            return None, 0, None, 0
        if self.infer_measured:
            return self._ab_in(l, True) + self._ab_out(l, True)
        return None, 0, self._ovf.data_ptr(), -1

    def _conv_fwd(self, x, l, bias, z, z_ld, stats, sc, sh, relu, N, h, w, st, bnin=None, measured=False, splitk=None, stats_rows=0):
        """bnin = (scale_ptr, shift_ptr): x is the producer's pre-BN tensor, BN + ReLU are applied on load.
        measured: inference -- the input bound is a measured one (8 replicas) and the epilogue folds max |output| per
        channel into the output's replicas (the next layer's range-guard bound)."""
        if measured:
            ab_in, ab_in_ld, ab_out, ab_out_ld = self._ab_infer(l)
        else:
            ab_in, ab_in_ld, ab_out, ab_out_ld = self._ab_in(l), 0, None, 0
        if bnin is not None:
            self.L.dc_conv3x3_fwd_bnin_f16x3(x, bnin[0], bnin[1], self._ab_in(l), _ptr(self.wp_fwd[l.name]), bias, z, z_ld,
                                             stats, stats_rows, sc, sh, relu, splitk, N, h, w, l.cin, l.cout, st)
        elif self.mfma == 'f16x3':
            self.L.dc_conv3x3_fwd_f16x3(x, _ptr(self.wp_fwd[l.name]), bias, z, z_ld, stats, stats_rows, sc, sh, relu, ab_in, ab_in_ld,
                                        ab_out, ab_out_ld, splitk, N, h, w, l.cin, l.cout, st)
        else:
            self.L.dc_conv3x3_fwd(x, _ptr(self.wp_fwd[l.name]), bias, z, z_ld, stats, sc, sh, relu,
                                  N, h, w, l.cin, l.cout, st)

    def _convT_fwd(self, x, l, bias, z, z_ld, stats, sc, sh, relu, N, h, w, st, bnin=None, measured=False):
        if measured:
            ab_in, ab_in_ld, ab_out, ab_out_ld = self._ab_infer(l)
        else:
            ab_in, ab_in_ld, ab_out, ab_out_ld = self._ab_in(l), 0, None, 0
        if bnin is not None:
            self.L.dc_convT2x2_fwd_bnin_f16x3(x, bnin[0], bnin[1], self._ab_in(l), _ptr(self.wp_fwd[l.name]), bias, z, z_ld,
                                              stats, sc, sh, relu, N, h, w, l.cin, l.cout, st)
        elif self.mfma == 'f16x3':
            self.L.dc_convT2x2_fwd_f16x3(x, _ptr(self.wp_fwd[l.name]), bias, z, z_ld, stats, sc, sh, relu, ab_in, ab_in_ld,
                                         ab_out, ab_out_ld, N, h, w, l.cin, l.cout, st)
        else:
            self.L.dc_convT2x2_fwd(x, _ptr(self.wp_fwd[l.name]), bias, z, z_ld, stats, sc, sh, relu,
                                   N, h, w, l.cin, l.cout, st)

    def _ab_key(self, l, out):
        """(array name, channel offset) of layer l's output (out=True) or input (out=False) bound."""
        n = l.name
        if out:
            if l.kind == 'convT':
                return 'cat%d' % l.lvl, 0                                 # up half of the concat
            if n[0] == 'e' and n[-1] == 'b':
                return 'cat%d' % l.lvl, self._cup(l.lvl)                  # skip half
            return n, 0
        if l.kind == 'convT':
            return ('bb' if l.lvl == 3 else 'd%db' % (l.lvl + 1)), 0
        if n[0] == 'd' and n[-1] == 'a':
            return 'cat%d' % l.lvl, 0
        if n[-1] == 'b':
            return n[:-1] + 'a', 0
        return 'cat%d' % (l.lvl - 1), self._cup(l.lvl - 1)                # e<lvl>a / ba: the pooled skip

    def _ab_out(self, l, measured=False):
        """Where layer l's activation bound goes: pointer, or (pointer, replica stride) for a measured (inference)
        bound.  None for the fp32-MFMA engine (no fp16 operands) / DC_RANGE_GUARD=0."""
        if self.mfma != 'f16x3' or not self.range_guard:
            return (None, 0) if measured else None
        key, off = self._ab_key(l, True)
        ptr = _ptr(self.abound, self._ab_off[key] + off)
        return (ptr, self._ab_ld[key]) if measured else ptr

    def _ab_in(self, l, measured=False):
        """Bound array of layer l's INPUT tensor, in l's input-channel order (None: the image / fp32-MFMA engine)."""
        if self.mfma != 'f16x3' or not self.range_guard or (l.kind == 'conv' and l.cin == 1):
            return (None, 0) if measured else None
        key, off = self._ab_key(l, False)
        ptr = _ptr(self.abound, self._ab_off[key] + off)
        return (ptr, self._ab_ld[key]) if measured else ptr

    def _ab_up(self, lvl, measured=False):
        """UpSampling2D branch: (bound of the up-sampled layer, its replica stride, up half of level lvl's concat bound,
        its replica stride); strides 0 in training (replica 0 only)."""
        if self.mfma != 'f16x3' or not self.range_guard:
            return None, 0, None, 0
        src, dst = 'bb' if lvl == 3 else 'd%db' % (lvl + 1), 'cat%d' % lvl
        return (_ptr(self.abound, self._ab_off[src]), self._ab_ld[src] if measured else 0,
                _ptr(self.abound, self._ab_off[dst]), self._ab_ld[dst] if measured else 0)

    def _bnin_src(self, prod, T):
        """(pre-BN tensor ptr, (scale_ptr, shift_ptr)) if the producer's activation is not materialised, else None."""
        if prod is None or prod.name not in self.nm:
            return None
        return _ptr(T['z_' + prod.name]), (self.stat_ptr(prod, 4), self.stat_ptr(prod, 5))

    def activation(self, name, N):
        """Post-BN/ReLU/dropout activation of layer `name` from the last forward_train as a torch tensor (debugging /
        tests): the stored tensor, or -- for a non-materialised layer -- rebuilt from z with torch ops."""
        l = self.by_name[name]
        A = self._acts(N)
        if name in self.nm:
            z = self._train_bufs(N)['z_' + name]
            o = self._stat_off[name]
            sc, sh = self.bstat[o + 4 * l.cout:o + 5 * l.cout], self.bstat[o + 5 * l.cout:o + 6 * l.cout]
            return torch.relu(torch.addcmul(sh, z, sc))
        c = l.cout
        if name.startswith('u'):
            return A['cat%d' % l.lvl][..., :c]
        if name.startswith('e') and name.endswith('b'):
            return A['cat%d' % l.lvl][..., self._cup(l.lvl):]
        return A[name]

    # ---- geometry helpers --------------------------------------------------------------------------------
    def _hw(self, lvl):
        return self.H >> lvl, self.W >> lvl

    def _cup(self, lvl):
        """Channels of the up-path half of level lvl's concat buffer (conv-transpose: c, UpSampling2D: 2c)."""
        return (2 if self.upsampling else 1) * (self.nfb << lvl)

    def _acts(self, N):
        """Activation tensors for batch size N (allocated once per N)."""
        key = ('acts', N)
        A = self._bufs.get(key)
        if A is not None:
            return A
        A = {}
        nfb = self.nfb
        dev = self.device

        def new(*shape, dtype=torch.float32):
            return torch.empty(shape, dtype=dtype, device=dev)

        for lvl in range(5):
            h, w = self._hw(lvl)
            c = nfb << lvl
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            A[tag + 'a'] = new(N, h, w, c)
            if lvl < 4:
                A['cat%d' % lvl] = new(N, h, w, self._cup(lvl) + c)   # [up path | skip]  (unet_2d_summary.py:200)
                A['pool%d' % lvl] = new(N, h // 2, w // 2, c)
                A['idx%d' % lvl] = new(N, h // 2, w // 2, c, dtype=torch.uint8)
                A['d%da' % lvl] = new(N, h, w, c)
                A['d%db' % lvl] = new(N, h, w, c)
            else:
                A['bb'] = new(N, h, w, c)
        A['p'] = new(N, self.H, self.W)
        # split-K slabs of the narrow (16^2 / 8^2) conv layers of a small-window forward -- inference, validation stripes, predict():
        # these run on the caller's stream like the training forward, which has its own copy in _train_bufs
        sk = 0
        if self.mfma == 'f16x3':
            for l in self.layers:
                if l.kind == 'conv' and l.cin > 1:
                    sk = max(sk, self.L.dc_conv3x3_splitk_ws_floats(N, *self._hw(l.lvl), l.cin, l.cout, 0))
        A['splitk_ws'] = new(sk) if sk > 0 else None
        self._bufs[key] = A
        return A

    def _plan(self, A):
        """('block', layer, input tensor, output tensor, output channel offset, output ld, h, w, producer layer of the
        input or None) / ('pool', ...) / ('up', ...) in execution order."""
        plan = []
        nfb = self.nfb
        prev, prev_l = None, None
        for lvl in range(5):
            h, w = self._hw(lvl)
            c = nfb << lvl
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            la, lb = self.by_name[tag + 'a'], self.by_name[tag + 'b']
            plan.append(('block', la, prev, A[tag + 'a'], 0, c, h, w, None))
            if lvl < 4:
                cup = self._cup(lvl)
                plan.append(('block', lb, A[tag + 'a'], A['cat%d' % lvl], cup, cup + c, h, w, la))
                plan.append(('pool', lvl, A['cat%d' % lvl], cup, cup + c, h, w))
                prev = A['pool%d' % lvl]
            else:
                plan.append(('block', lb, A[tag + 'a'], A['bb'], 0, c, h, w, la))
                prev, prev_l = A['bb'], lb
        for lvl in (3, 2, 1, 0):
            h, w = self._hw(lvl)
            c = nfb << lvl
            la, lb = self.by_name['d%da' % lvl], self.by_name['d%db' % lvl]
            if self.upsampling:
                plan.append(('up', lvl, prev, A['cat%d' % lvl], 3 * c, h, w))    # h,w are OUTPUT dims
            else:
                plan.append(('block', self.by_name['u%d' % lvl], prev, A['cat%d' % lvl], 0, 2 * c, h, w, prev_l))   # convT: OUTPUT dims
            plan.append(('block', la, A['cat%d' % lvl], A['d%da' % lvl], 0, c, h, w, None))
            plan.append(('block', lb, A['d%da' % lvl], A['d%db' % lvl], 0, c, h, w, la))
            prev, prev_l = A['d%db' % lvl], lb
        return plan

    # ---- inference ---------------------------------------------------------------------------------------
    def _check_input(self, name, t, dtype, shape=None):
        """Raw pointers cross the C ABI: a wrong dtype / layout / device would silently compute garbage."""
        shape = tuple(shape) if shape is not None else (t.shape[0], self.H, self.W)
        if not isinstance(t, torch.Tensor) or t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous() \
                or t.device != self.device:
            raise ValueError('%s must be a contiguous %s tensor of shape %r on %s, got %s' % (
                name, dtype, shape, self.device,
                '%s %r on %s%s' % (t.dtype, tuple(t.shape), t.device, '' if t.is_contiguous() else ' (non-contiguous)')
                if isinstance(t, torch.Tensor) else type(t).__name__))

    @_on_device
    def forward_infer(self, x_dev):
        """x_dev: float32 cuda tensor (N,H,W) -> p: (N,H,W) probabilities (BN folded into the conv epilogue).  A steady-state
        forward is a fixed launch sequence: recorded once per (batch, mode, stream) and replayed from C like a train step (_taped)."""
        N = x_dev.shape[0]
        self._check_input('x', x_dev, torch.float32)
        A = self._acts(N)
        key = ('infer', N, self._stream(), bool(self.infer_measured), self._packed_dirty, self._fold_dirty)
        return self._taped(key, lambda: self._forward_infer_body(x_dev, N, A), lambda: {'x': _ptr(x_dev)},
                           post_attrs=('_packed_dirty', '_fold_dirty'))

    def _forward_infer_body(self, x_dev, N, A):
        L, st = self.L, self._stream()
        xp = self._v('x', _ptr(x_dev))
        self.repack()
        self.refold()
        if self.mfma == 'f16x3' and self.range_guard:
            if self.infer_measured:
                L.dc_fill(self.abound.data_ptr(), self.abound.numel(), 0.0, st)   # the conv epilogues fold the measured max |a| per channel into it
            else:
                L.dc_fill(self._ovf.data_ptr(), self._ovf.numel(), 0.0, st)       # optimistic: they only raise this flag (forward_infer_checked looks at it)
        plan = self._plan(A)
        sk = _ptr(A['splitk_ws']) if A.get('splitk_ws') is not None else None
        pooled_by_conv = False
        for si, step in enumerate(plan):
            if step[0] == 'pool':
                if pooled_by_conv:             # the convolution's epilogue has written the pooled tensor
                    pooled_by_conv = False
                    continue
                _, lvl, src, coff, ld, h, w = step
                L.dc_maxpool2x2_fwd(_ptr(src, coff), ld, _ptr(A['pool%d' % lvl]), None, N, h, w, self.nfb << lvl, st)
                continue
            if step[0] == 'up':
                _, lvl, src, dst, ld, h, w = step
                L.dc_upsample2x_drop_fwd(_ptr(src), _ptr(dst), ld, None, 1.0, 0,
                                         *(self._ab_up(lvl, True) if self.infer_measured else (None, 0, None, 0)),
                                         N, h // 2, w // 2, self._cup(lvl), st)
                continue
            _, l, src, dst, coff, ld, h, w, _prod = step
            sc, sh = self.stat_ptr(l, 2), self.stat_ptr(l, 3)
            if l.kind == 'conv' and l.cin == 1:
                L.dc_conv3x3_c1_fwd(xp, self.pview(self.pflat, l, 'k'), None, _ptr(dst, coff), ld, None,
                                    sc, sh, 1, *self._ab_infer(l)[2:], N, h, w, l.cout, st)
            elif l.kind == 'conv':
                nxt = plan[si + 1] if si + 1 < len(plan) else None
                if (self.mfma == 'f16x3' and not (self.range_guard and self.infer_measured)
                        and nxt is not None and nxt[0] == 'pool' and nxt[2] is dst and nxt[3] == coff
                        and L.dc_conv3x3_fwd_pool_blocks(N, h, w, l.cin, l.cout) > 0):
                    # the block in front of a max-pool: activation (into the concat buffer) and pooled tensor in one kernel
                    flag = self._ovf.data_ptr() if self.range_guard else None
                    L.dc_conv3x3_fwd_pool_f16x3(_ptr(src), _ptr(self.wp_fwd[l.name]), None, _ptr(dst, coff), ld, sc, sh, 1, flag,
                                                _ptr(A['pool%d' % nxt[1]]), N, h, w, l.cin, l.cout, st)
                    pooled_by_conv = True
                    continue
                self._conv_fwd(_ptr(src), l, None, _ptr(dst, coff), ld, None, sc, sh, 1, N, h, w, st, measured=True, splitk=sk)
            else:
                self._convT_fwd(_ptr(src), l, None, _ptr(dst, coff), ld, None, sc, sh, 1, N, h // 2, w // 2, st, measured=True)
        lo = self.by_name['out']
        L.dc_head_fwd(_ptr(A['d0b']), self.pview(self.pflat, lo, 'k'), self.pview(self.pflat, lo, 'b'), None,
                      _ptr(A['p']), None, N * self.H * self.W, self.nfb, st)
        return A['p']

    def forward_infer_checked(self, x_dev):
        """forward_infer + the range check of the optimistic mode: if an activation left fp16's range the pass is
        repeated with measured per-channel bounds (and the engine stays in that mode).  Synchronises the stream."""
        p = self.forward_infer(x_dev)
        if self.mfma == 'f16x3' and self.range_guard and not self.infer_measured and float(self._ovf[0].item()) != 0.0:
            self.infer_measured = True
            p = self.forward_infer(x_dev)
        return p

    @_on_device
    def predict_tta(self, img, augmentations, hs, ws, threshold, return_mean=False):
        """predict(augmentation=True) for ONE padded (H,W) float32 numpy image, on the device: the K augmented copies are
        gathered by dc_gather_maps, go through one batch-K forward, and dc_tta_merge inverse-maps, averages (double,
        table order), crops to (hs, ws) and thresholds.  augmentations = [(name, fwd, inv)] on (N,H,W) arrays.
        Returns the uint8 mask (hs, ws) -- with return_mean also the float32 mean probability map (hs, ws) it thresholds.
        Host traffic: 4*H*W bytes in, hs*ws bytes out (vs 8x both ways)."""
        K, H, W = len(augmentations), self.H, self.W
        n = H * W
        key = ('tta', K, tuple(name for name, _, _ in augmentations))
        m = self._bufs.get(key)
        if m is None:
            idx = np.arange(n, dtype=np.int32).reshape(1, H, W)
            fwd = np.stack([np.ascontiguousarray(f(idx)).reshape(-1) for _, f, _ in augmentations])
            inv = np.stack([np.ascontiguousarray(g(idx)).reshape(-1) for _, _, g in augmentations])
            if fwd.shape != (K, n) or inv.shape != (K, n):
                raise ValueError('test-time augmentations must map (N,H,W) to (N,H,W) on a square window')
            # out_k[p] = in[fwd_k[p]];  inv_k(pred_k)[p] = pred_k[inv_k[p]]
            m = dict(fwd=torch.from_numpy(fwd).to(self.device), inv=torch.from_numpy(inv).to(self.device),
                     src=torch.empty(n, dtype=torch.float32, device=self.device),
                     x=torch.empty((K, H, W), dtype=torch.float32, device=self.device),
                     mask=torch.empty(n, dtype=torch.uint8, device=self.device),
                     host=_pinned('tta1_img', (H, W), torch.float32), mask_host=_pinned('tta1_mask', (n,), torch.uint8))
            self._bufs[key] = m
        L, st = self.L, self._stream()
        m['host'].numpy()[...] = img
        m['src'].copy_(m['host'].view(-1), non_blocking=True)
        L.dc_gather_maps(_ptr(m['src']), m['fwd'].data_ptr(), _ptr(m['x']), K, n, st)
        p = self.forward_infer_checked(m['x'])
        mean = torch.empty(hs * ws, dtype=torch.float32, device=self.device) if return_mean else None
        L.dc_tta_merge(_ptr(p), m['inv'].data_ptr(), K, H, W, int(hs), int(ws), float(threshold), m['mask'].data_ptr(),
                       _ptr(mean) if return_mean else None, st)
        m['mask_host'][:hs * ws].copy_(m['mask'][:hs * ws], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        mask = m['mask_host'][:hs * ws].numpy().reshape(hs, ws).copy()
        return (mask, mean.cpu().numpy().reshape(hs, ws)) if return_mean else mask

    def tta_begin(self, count, augmentations):
        """Pipelined predict() over `count` datasets: see _TtaJob."""
        return _TtaJob(self, count, augmentations)

    # ---- training ----------------------------------------------------------------------------------------
    def _train_bufs(self, N):
        key = ('train', N)
        T = self._bufs.get(key)
        if T is not None:
            return T
        L = self.L
        T = {}
        dev = self.device
        nfb = self.nfb
        ws_floats, stats_floats, part_floats = 0, 0, 0
        for l in self.layers:
            if l.kind == 'head':
                continue
            h, w = self._hw(l.lvl)
            T['z_' + l.name] = torch.empty((N, h, w, l.cout), dtype=torch.float32, device=dev)
            if l.kind == 'conv':
                tiles = L.dc_conv3x3_c1_tiles(N, h, w, l.cout) if l.cin == 1 else L.dc_conv3x3_tiles(N, h, w, l.cout)
                stats_floats = max(stats_floats, tiles * l.cout * 2)
                ws_floats = max(ws_floats, L.dc_conv3x3_wgrad_ws_floats(N, h, w, l.cin, l.cout))
            else:
                tiles = (L.dc_convT2x2_f16x3_tiles if self.mfma == 'f16x3' else L.dc_convT2x2_tiles)(N, h // 2, w // 2, l.cout)
                stats_floats = max(stats_floats, tiles * 4 * l.cout * 2)
                ws_floats = max(ws_floats, L.dc_convT2x2_wgrad_ws_floats(N, h // 2, w // 2, l.cin, l.cout))
            blocks = L.dc_bn_bwd_blocks(N * h * w, l.cout)
            part_floats = max(part_floats, blocks * l.cout * 2)
        hb = L.dc_head_blocks(N * self.H * self.W)
        part_floats = max(part_floats, hb * (nfb + 4), hb * 12, hb * nfb * 2)
        for l in self.layers:            # data gradients that emit the previous layer's BatchNorm-backward sums
            if l.kind == 'conv' and l.cin > 1:
                h, w = self._hw(l.lvl)
                part_floats = max(part_floats, L.dc_conv3x3_dgrad_bnred_blocks(N, h, w, l.cin, l.cout) * l.cin * 2)
        for lvl in range(4):
            h, w = self._hw(lvl)
            part_floats = max(part_floats, L.dc_maxpool2x2_bwd_blocks(N, h, w, nfb << lvl) * (nfb << lvl) * 2)
        for l in self.layers:
            if l.kind == 'conv':
                part_floats = max(part_floats, L.dc_conv3x3_bwd_joint_blocks(N, *self._hw(l.lvl), l.cin, l.cout) * l.cin * 2)
            elif l.kind == 'convT':
                h, w = self._hw(l.lvl)
                part_floats = max(part_floats, L.dc_convT2x2_dgrad_bnred_blocks(N, h // 2, w // 2, l.cin, l.cout) * l.cin * 2)
        # split-K slabs of the narrow conv layers (forward and data gradients all run on the caller's stream: one workspace)
        sk = 4
        for l in self.layers:
            if l.kind == 'conv' and l.cin > 1 and self.mfma == 'f16x3':
                h, w = self._hw(l.lvl)
                sk = max(sk, L.dc_conv3x3_splitk_ws_floats(N, h, w, l.cin, l.cout, 0), L.dc_conv3x3_splitk_ws_floats(N, h, w, l.cin, l.cout, 1))
            elif l.kind == 'convT' and self.mfma == 'f16x3':
                h, w = self._hw(l.lvl)
                sk = max(sk, L.dc_convT2x2_dgrad_splitk_ws_floats(N, h // 2, w // 2, l.cin, l.cout))
        T['splitk_ws'] = torch.empty(sk, dtype=torch.float32, device=dev)
        T['stats_ws'] = torch.empty(stats_floats, dtype=torch.float64, device=dev)      # (sum, sum of squares) partials
        # partial rows folded to FOLD_ROWS before the finalize where a launch leaves thousands of them (dc_bn_stats_rows_fold)
        T['stats_fold'] = torch.empty(self.FOLD_ROWS * 4 * max(l.cout for l in self.layers) * 2, dtype=torch.float64, device=dev)
        T['part_ws'] = torch.empty(part_floats, dtype=torch.float32, device=dev)
        T['part_ws2'] = torch.empty(part_floats, dtype=torch.float32, device=dev)
        # per-(row, channel) max |dy| next to the pass-1 sums of the same producer (dc_bn_bwd_finalize_dzin)
        T['amax_ws'] = torch.empty(part_floats // 2 + 4, dtype=torch.float32, device=dev)
        T['amax_ws2'] = torch.empty(part_floats // 2 + 4, dtype=torch.float32, device=dev)
        T['head_gpart'] = torch.empty(hb * (nfb + 4), dtype=torch.float32, device=dev)
        # per dz buffer: the apply pass' per-block max |dz| and conv-bias-gradient partials (read by the finalize launch on the
        # weight-gradient stream and by the data gradient: they rotate with the dz buffer they describe)
        big = N * self.H * self.W * nfb
        S = self.SLOTS_DEEP if (self.deep_slots and 2 * self.SLOTS_DEEP * big * 4 <= self.SLOTS_DEEP_MAX_BYTES) else self.SLOTS
        T['slots'] = S
        T['absmax'] = [torch.empty(4096, dtype=torch.float32, device=dev) for _ in range(S)]
        dpart = max(L.dc_bn_bwd_blocks(N * self._hw(l.lvl)[0] * self._hw(l.lvl)[1], l.cout) * l.cout
                    for l in self.layers if l.kind != 'head')
        T['dbias_part'] = [torch.empty(dpart, dtype=torch.float32, device=dev) for _ in range(S)]
        T['dz_scale'] = torch.ones(S * 4, dtype=torch.float32, device=dev)      # one (16-B aligned) scalar per dz buffer
        cmax = max(l.cout for l in self.layers if l.kind != 'head')
        T['dz_coef'] = [torch.zeros(7 * cmax, dtype=torch.float32, device=dev) for _ in range(S)]   # dz-on-load tables
        T['red_tmp'] = torch.empty(32 * 1024, dtype=torch.float32, device=dev)
        T['wgrad_ws'] = torch.empty(ws_floats, dtype=torch.float32, device=dev)
        # the first layer's weight gradient may run on the MAIN stream beside the side stream's last one (tail_on_main): its own slabs
        l0 = self.layers[0]
        T['wgrad_ws_main'] = torch.empty(max(L.dc_conv3x3_wgrad_ws_floats(N, self.H, self.W, l0.cin, l0.cout), 4), dtype=torch.float32,
                                         device=dev)
        # the joint data- + weight-gradient kernel of the 32 -> 32 blocks runs on the MAIN stream: its own slab workspace
        # (the side stream's weight gradients share wgrad_ws)
        jws = max([L.dc_conv3x3_bwd_joint_ws_floats(N, *self._hw(l.lvl), l.cin, l.cout) for l in self.layers if l.kind == 'conv'] + [4])
        T['joint_ws'] = torch.empty(jws, dtype=torch.float32, device=dev)
        T['sums'] = torch.zeros(12, dtype=torch.float64, device=dev)
        T['bn_sums'] = torch.zeros(2 * max(l.cout for l in self.layers if l.kind != 'head'), dtype=torch.float64, device=dev)
        T['bn_gsum'] = torch.zeros(2 * max(l.cout for l in self.layers if l.kind != 'head'), dtype=torch.float32, device=dev)
        big = N * self.H * self.W * nfb
        # dz rotates over 3 buffers: the weight-gradient kernels run on a side stream and may still be reading the
        # dz of block L while the main stream already produces the dz of block L-1 / L-2.  The activation gradients
        # rotate over 3 buffers for the same reason: with dz formed on load the weight gradient of block L reads `da`.
        T['dz'] = [torch.empty(big, dtype=torch.float32, device=dev) for _ in range(S)]
        T['g'] = [torch.empty(big, dtype=torch.float32, device=dev) for _ in range(S)]
        for lvl in range(4):
            h, w = self._hw(lvl)
            T['dcat%d' % lvl] = torch.empty(N * h * w * (self._cup(lvl) + (nfb << lvl)), dtype=torch.float32, device=dev)
        self._bufs[key] = T
        return T

    # ---- training batches cut on the device (dc_crop_augment) -------------------------------------------------------
    def set_crop_sources(self, S_list, M_list):
        """Upload the datasets' normalised summary images (float32) and flattened masks (uint8) ONCE, concatenated in
        model.crop_layout()'s order: what UNet2DSummary._batch_gen crops on the host per step (unet_2d_summary.py:505-521)
        is cut out of these by dc_crop_augment from then on."""
        from .model import crop_layout
        shapes = [tuple(np.shape(v)) for v in S_list]
        if [tuple(np.shape(v)) for v in M_list] != shapes or any(len(sh) != 2 for sh in shapes):
            raise ValueError('summaries and masks must be 2-D arrays of equal shapes per dataset')
        offs, total = crop_layout(shapes)
        S, M = np.zeros(total, np.float32), np.zeros(total, np.uint8)
        for o, (h, w), sv, mv in zip(offs, shapes, S_list, M_list):
            S[o:o + h * w] = np.asarray(sv, dtype=np.float32).ravel()
            M[o:o + h * w] = np.asarray(mv).astype(np.uint8).ravel()
        self._crop = dict(S=torch.from_numpy(S).to(self.device), M=torch.from_numpy(M).to(self.device), total=total)

    @_on_device
    def crop_batch(self, items):
        """items: int64 (n, 4) host array (model.DeviceBatch.items) -> (x float32 (n,H,W), y uint8 (n,H,W)) device tensors of
        this engine, written by dc_crop_augment on the current stream (the step's input buffers: the launch is ordered
        behind the previous step's last reader of them, the first layer's weight gradient, because adam_step() joins it)."""
        src = getattr(self, '_crop', None)
        if src is None:
            raise RuntimeError('set_crop_sources() first')
        if self.H != self.W:
            raise ValueError('device-side batches need a square window')
        items = np.ascontiguousarray(items, dtype=np.int64)
        n = int(items.shape[0])
        key = ('xy_in', n)
        xy = self._bufs.get(key)
        if xy is None:
            xy = self._bufs[key] = (torch.empty((n, self.H, self.W), dtype=torch.float32, device=self.device),
                                    torch.empty((n, self.H, self.W), dtype=torch.uint8, device=self.device))
        self._settle_tail()
        self.L.dc_crop_augment(src['S'].data_ptr(), src['M'].data_ptr(), src['total'], items.ctypes.data, n, self.H,
                               xy[0].data_ptr(), xy[1].data_ptr(), self._stream())
        return xy

    def _drop_args(self, l, masks, step_seed):
        if l.drop <= 0.0:
            return None, 1.0, 0
        keep = 1.0 - l.drop
        if masks is not None:
            m = masks[l.name]
            assert m.dtype == torch.uint8 and m.is_contiguous()
            return m.data_ptr(), keep, 0
        h, w = self._hw(l.lvl)
        seed = (step_seed * 1000003 + l.index * 7919) & 0xFFFFFFFFFFFFFFFF
        return None, keep, self._v(('seed', l.name), parallel.shard_drop_seed(seed, self._last[0] * h * w * l.cout))

    def _up_drop_args(self, lvl, masks, step_seed):
        rate = self.up_drop[lvl]
        if rate <= 0.0:
            return None, 1.0, 0
        if masks is not None:
            m = masks['u%d' % lvl]
            assert m.dtype == torch.uint8 and m.is_contiguous()
            return m.data_ptr(), 1.0 - rate, 0
        h, w = self._hw(lvl)
        seed = (step_seed * 1000003 + (100 + lvl) * 7919) & 0xFFFFFFFFFFFFFFFF
        return None, 1.0 - rate, self._v(('useed', lvl), parallel.shard_drop_seed(seed, self._last[0] * h * w * self._cup(lvl)))

    @_on_device
    def forward_train(self, x_dev, y_dev, masks=None, update_moving=True):
        """Training-mode forward (batch statistics, dropout).  Returns p; loss/metric sums land in T['sums']."""
        N = x_dev.shape[0]
        self._check_input('x', x_dev, torch.float32)
        self._check_input('y', y_dev, torch.uint8, (N, self.H, self.W))
        if masks is not None:
            for l in self.layers:
                if l.drop > 0.0:
                    h, w = self._hw(l.lvl)
                    self._check_input('masks[%r]' % l.name, masks.get(l.name), torch.uint8, (N, h, w, l.cout))
            if self.upsampling:
                for lvl, rate in self.up_drop.items():
                    if rate > 0.0:
                        h, w = self._hw(lvl)
                        self._check_input("masks['u%d']" % lvl, masks.get('u%d' % lvl), torch.uint8, (N, h, w, self._cup(lvl)))
        self._settle_tail()
        self._acts(N), self._train_bufs(N)
        step_seed = self.drop_seed + self.iterations
        self._last = (N, masks, step_seed, x_dev, y_dev)
        world = parallel.world_size()
        sync = self._sync_bn()

        def body():
            return self._forward_train_body(x_dev, y_dev, masks, update_moving)
        if masks is not None or sync:          # explicit mask pointers / collectives between the launches: not taped
            return body()
        key = ('fwd', N, bool(update_moving), self.loss_kind, self._stream(), self._packed_dirty, world, parallel.rank())
        return self._taped(key, body, lambda: self._step_values(x_dev, y_dev, step_seed),
                           post_attrs=('_packed_dirty', '_fold_dirty', '_head_bwd_done'))

    def _step_values(self, x_dev, y_dev, step_seed):
        """This step's value of every per-step launch argument (_v keys) of the forward / backward tapes."""
        vals = {'x': _ptr(x_dev), 'y': y_dev.data_ptr()}
        for l in self.layers:
            if l.drop > 0.0:
                vals[('seed', l.name)] = self._drop_args(l, None, step_seed)[2]
        if self.upsampling:
            for lvl, rate in self.up_drop.items():
                if rate > 0.0:
                    vals[('useed', lvl)] = self._up_drop_args(lvl, None, step_seed)[2]
        return vals

    def _forward_train_body(self, x_dev, y_dev, masks, update_moving):
        N = x_dev.shape[0]
        L, st = self.L, self._stream()
        self.repack()
        A, T = self._acts(N), self._train_bufs(N)
        step_seed = self._last[2]
        world = parallel.world_size()
        sync = self._sync_bn()
        xp, yp = self._v('x', _ptr(x_dev)), self._v('y', y_dev.data_ptr())
        plan = self._plan(A)
        pooled_with_block = False
        for si, step in enumerate(plan):
            if step[0] == 'pool':
                if pooled_with_block:          # the block's BN + ReLU + dropout pass has pooled it already
                    pooled_with_block = False
                    continue
                _, lvl, src, coff, ld, h, w = step
                L.dc_maxpool2x2_fwd(_ptr(src, coff), ld, _ptr(A['pool%d' % lvl]), A['idx%d' % lvl].data_ptr(),
                                    N, h, w, self.nfb << lvl, st)
                continue
            if step[0] == 'up':
                _, lvl, src, dst, ld, h, w = step
                mptr, keep, seed = self._up_drop_args(lvl, masks, step_seed)
                L.dc_upsample2x_drop_fwd(_ptr(src), _ptr(dst), ld, mptr, keep, seed, *self._ab_up(lvl), N, h // 2, w // 2,
                                         self._cup(lvl), st)
                continue
            _, l, src, dst, coff, ld, h, w, prod = step
            z = T['z_' + l.name]
            bias = self.pview(self.pflat, l, 'b')
            stats = T['stats_ws'].data_ptr()
            groups = 1
            bsrc = self._bnin_src(prod, T)
            xin, bn = (bsrc[0], bsrc[1]) if bsrc is not None else (_ptr(src) if src is not None else None, None)
            if l.kind == 'conv' and l.cin == 1:
                tiles = L.dc_conv3x3_c1_tiles(N, h, w, l.cout)
                L.dc_conv3x3_c1_fwd(xp, self.pview(self.pflat, l, 'k'), bias, _ptr(z), l.cout, stats,
                                    None, None, 0, None, 0, N, h, w, l.cout, st)
            elif l.kind == 'conv':
                # BatchNorm-partial rows: one per (workgroup, consumer set) where the role-split kernel serves the launch (<= 512
                # rows for the finalize launch instead of up to 8 192 tiles), one per pixel tile elsewhere
                tiles = (L.dc_conv3x3_stats_rows(N, h, w, l.cin, l.cout) if self.mfma == 'f16x3' and self.stats_per_wg
                         else L.dc_conv3x3_tiles(N, h, w, l.cout))
                self._conv_fwd(xin, l, bias, _ptr(z), l.cout, stats, None, None, 0, N, h, w, st, bnin=bn, splitk=_ptr(T['splitk_ws']),
                               stats_rows=tiles)
            else:
                tiles = (L.dc_convT2x2_f16x3_tiles if self.mfma == 'f16x3' else L.dc_convT2x2_tiles)(N, h // 2, w // 2, l.cout)
                groups = 4
                self._convT_fwd(xin, l, bias, _ptr(z), l.cout, stats, None, None, 0, N, h // 2, w // 2, st, bnin=bn)
            pixels = N * h * w
            mom = l.mom if update_moving else -1.0
            if self.FOLD_MIN > 0 and tiles * groups >= self.FOLD_MIN and tiles >= 2:
                fold, chunks = T['stats_fold'].data_ptr(), min(self.FOLD_ROWS, tiles // 2)
                L.dc_bn_stats_rows_fold(stats, tiles, groups, l.cout, chunks, fold, st)
                stats, tiles = fold, chunks
            if sync:
                # 'sync' BatchNorm: per-channel (sum, sum of squares) -> all-reduce over the ranks -> statistics
                bs = T['bn_sums'][:2 * l.cout]
                L.dc_bn_stats_reduce(stats, tiles, groups, l.cout, bs.data_ptr(), st)
                self._sync_all_reduce(bs)
                nm_l = l.name in self.nm
                L.dc_bn_stats_finalize_sums(bs.data_ptr(), l.cout, float(world * pixels), BN_EPS, mom,
                                            self.stat_ptr(l, 0), self.stat_ptr(l, 1), self.sview(l, 'mmean'),
                                            self.sview(l, 'mvar'), self.pview(self.pflat, l, 'gamma'),
                                            self.pview(self.pflat, l, 'beta'), self.stat_ptr(l, 4) if nm_l else None,
                                            self.stat_ptr(l, 5) if nm_l else None, self._ab_out(l) if nm_l else None, st)
                if nm_l:
                    continue
            elif l.name in self.nm:
                # activation not materialised: emit the per-channel affine its consumers apply on load
                L.dc_bn_stats_finalize_affine(stats, tiles, groups, l.cout, float(pixels), BN_EPS, mom,
                                              self.stat_ptr(l, 0), self.stat_ptr(l, 1), self.sview(l, 'mmean'),
                                              self.sview(l, 'mvar'), self.pview(self.pflat, l, 'gamma'),
                                              self.pview(self.pflat, l, 'beta'), self.stat_ptr(l, 4),
                                              self.stat_ptr(l, 5), self._ab_out(l), st)
                continue
            if not sync:
                L.dc_bn_stats_finalize(stats, tiles, groups, l.cout, float(pixels), BN_EPS, mom,
                                       self.stat_ptr(l, 0), self.stat_ptr(l, 1), self.sview(l, 'mmean'),
                                       self.sview(l, 'mvar'), st)
            mptr, keep, seed = self._drop_args(l, masks, step_seed)
            nxt = plan[si + 1] if si + 1 < len(plan) else None
            if nxt is not None and nxt[0] == 'pool' and nxt[2] is dst and nxt[3] == coff:
                lvl = nxt[1]
                L.dc_bn_relu_drop_pool_fwd(_ptr(z), self.stat_ptr(l, 0), self.stat_ptr(l, 1),
                                           self.pview(self.pflat, l, 'gamma'), self.pview(self.pflat, l, 'beta'),
                                           mptr, keep, seed, _ptr(dst, coff), ld, _ptr(A['pool%d' % lvl]),
                                           A['idx%d' % lvl].data_ptr(), N, h, w, l.cout,
                                           float((world if sync else 1) * pixels), self._ab_out(l), st)
                pooled_with_block = True
                continue
            L.dc_bn_relu_drop_fwd(_ptr(z), self.stat_ptr(l, 0), self.stat_ptr(l, 1),
                                  self.pview(self.pflat, l, 'gamma'), self.pview(self.pflat, l, 'beta'),
                                  mptr, keep, seed, _ptr(dst, coff), ld, pixels, l.cout,
                                  float((world if sync else 1) * pixels), self._ab_out(l), st)
        if update_moving:
            self._fold_dirty = True
        lo = self.by_name['out']
        pixels = N * self.H * self.W
        hb = L.dc_head_blocks(pixels)
        hsrc = self._bnin_src(self.by_name['d0b'], T)
        self._head_bwd_done = False
        if self.loss_kind in (0, 1):
            # per-pixel losses: the head's backward rides on its forward (one pass over the 512^2 x nfb tensor instead of
            # two); backward() picks up da, the weight-gradient partials and d0b's BatchNorm-backward sums from here
            a_in, sc_in, sh_in = (hsrc[0], hsrc[1][0], hsrc[1][1]) if hsrc is not None else (_ptr(A['d0b']), None, None)
            ld0 = self.by_name['d0b']
            red = hsrc is not None
            L.dc_head_fwd_bwd(a_in, sc_in, sh_in, self.pview(self.pflat, lo, 'k'), self.pview(self.pflat, lo, 'b'),
                              yp, _ptr(A['p']), _ptr(T['part_ws']), _ptr(T['g'][0]), _ptr(T['head_gpart']),
                              self.loss_kind, self.stat_ptr(ld0, 0) if red else None, self.stat_ptr(ld0, 1) if red else None,
                              _ptr(T['part_ws2']) if red else None, _ptr(T['amax_ws2']) if red else None, pixels, self.nfb, st)
            self._head_bwd_done = True
        elif hsrc is not None:
            L.dc_head_fwd_bnin(hsrc[0], hsrc[1][0], hsrc[1][1], self.pview(self.pflat, lo, 'k'),
                               self.pview(self.pflat, lo, 'b'), yp, _ptr(A['p']), _ptr(T['part_ws']),
                               pixels, self.nfb, st)
        else:
            L.dc_head_fwd(_ptr(A['d0b']), self.pview(self.pflat, lo, 'k'), self.pview(self.pflat, lo, 'b'),
                          yp, _ptr(A['p']), _ptr(T['part_ws']), pixels, self.nfb, st)
        L.dc_reduce_partials_f64(_ptr(T['part_ws']), hb, 12, T['sums'].data_ptr(), st)
        return A['p']

    def grad_buckets(self):
        """gflat as three contiguous ranges in the order the backward completes them: decoder + head, bottleneck,
        encoder (flat layout = Keras get_weights order: encoder, bottleneck, decoder, head)."""
        o_ba, o_dec = self.by_name['ba'].off['k'][0], self.layers[10].off['k'][0]
        return [(o_dec, self.n_train), (o_ba, o_dec), (0, o_ba)]

    def _dzin_ok(self, l, N):
        """Does block l run without the BatchNorm-backward apply pass?  0: no (finalize -> apply -> gradients);
        1: dz formed on load by BOTH gradient kernels (level 0: joint / dz-on-load kernels; DC_DZIN=all: everywhere eligible);
        2 (round 5, levels >= 1): the data gradient forms dz on load AND writes it once, the weight gradient is the plain kernel --
           the dz-on-load weight gradient is the slower of the two (its producers are its longest role), the data gradient's
           producers have the values in registers anyway: 1 tensor write replaces the apply pass' 2 reads + 1 write."""
        if not self.dzin or l.kind != 'conv' or l.drop > 0.0:
            return 0
        h, w = self._hw(l.lvl)
        if not self.dzin_all and N * self.H * self.W <= (1 << 20):
            # small steps (the reference's own 96^2 / 128^2 training windows) are launch-latency bound, not HBM-bound: the
            # apply pass costs ~15 us there while the persistent dz-on-load kernels pay their set-up (same-box 3.29 vs 3.92 ms)
            return 0
        if l.lvl in self.dzin_lvls:
            if l.cin == 1:
                return 1
            return 1 if self.L.dc_conv3x3_dgrad_dzin_blocks(N, h, w, l.cin, l.cout) > 0 else 0
        if self.dz_writeback and l.cin > 32 and self.L.dc_conv3x3_dgrad_dzin_blocks(N, h, w, l.cin, l.cout) > 0:
            return 2          # (the write-back lives in the 64-column instantiation: layers with more than 32 input channels)
        return 0

    @_on_device
    def backward(self, bucket_cb=None, defer_tail=False, comm=None):
        """Backward of the last forward_train: fills gflat (same layout as pflat).  defer_tail (single-GPU training step
        only): return WITHOUT joining the weight-gradient stream -- the only thing still running there is the first layer's
        weight (and bias) gradient, and adam_step() updates everything else and re-packs the weights underneath it before
        it joins (the first layer's ~320 parameters follow).  bucket_cb(lo, hi), if given, is called
        with the side (weight-gradient) stream current as soon as gflat[lo:hi] is complete on it -- data-parallel training
        starts that range's all-reduce there, so it overlaps with the rest of the backward (grad_buckets()[:2]; the last
        range is complete when backward() returns).

        Per Conv -> BN -> ReLU block, in reverse order (unet_2d_summary.py:163-167):
          pass-1 sums (sum dy, sum dy*xhat [, max |dy|]) -- emitted by the kernel that WROTE da wherever one exists (head,
            max-pool backward, the data gradient above), else dc_bn_bwd_reduce;
          Dropout-free conv blocks: dc_bn_bwd_finalize_dzin -> data gradient and weight gradient form dz on load;
          the others (Dropout, conv-transpose, small shapes): finalize -> dc_bn_bwd_apply (writes dz) -> gradients.
        Two HIP streams: the critical path (sums -> data gradient -> next block) stays on the caller's stream, the weight
        gradients go to a side stream.  Hand-offs are stream events; what a weight gradient reads (dz or the table, and
        `da`) rotates over SLOTS buffers guarded by the event of the weight gradient that read it last."""
        N, masks, step_seed, x_dev, y_dev = self._last
        self._settle_tail()
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream if self.streams == 2 else torch.cuda.current_stream(self.device)
        world = parallel.world_size()
        sync = self._sync_bn()

        if comm is not None:
            if bucket_cb is not None:
                raise ValueError('backward: pass bucket_cb (the caller issues the collectives) OR comm (dc_comm_* from here), not both')

        def body():
            return self._backward_body(bucket_cb, defer_tail, comm)

        def on_mark(i):                      # replay: the same hand-over of a finished gradient range to the collective
            with torch.cuda.stream(side):
                bucket_cb(*self.grad_buckets()[i])
        if masks is not None or sync:
            return body()
        key = ('bwd', N, self.loss_kind, self._stream(), side.cuda_stream, self._head_bwd_done, bool(defer_tail),
               bucket_cb is not None, world, parallel.rank(), comm, self.ar_buckets)
        return self._taped(key, body, lambda: self._step_values(x_dev, y_dev, step_seed),
                           post_attrs=('_tail', '_head_bwd_done', '_ev_i', '_evf_i'), on_mark=on_mark)

    def _backward_body(self, bucket_cb, defer_tail, comm=None):
        N, masks, step_seed, x_dev, y_dev = self._last
        L, st = self.L, self._stream()
        A, T = self._acts(N), self._train_bufs(N)
        self._ev_i = 0
        self._evf_i = 0
        self._tail = None
        xp, yp = self._v('x', _ptr(x_dev)), self._v('y', y_dev.data_ptr())
        nfb = self.nfb
        f16 = self.mfma == 'f16x3'
        pixels0 = N * self.H * self.W
        lo = self.by_name['out']
        hb = L.dc_head_blocks(pixels0)
        hsrc = self._bnin_src(self.by_name['d0b'], T)
        gb = T['g']
        fused_d0b = None          # (partial ptr, amax ptr, rows): pass-1 sums of d0b already produced by the head kernel
        gpart = T['part_ws']
        if self._head_bwd_done:   # forward_train's fused head kernel has written g[0], the gradient partials and the sums
            self._head_bwd_done = False
            gpart = T['head_gpart']
            if hsrc is not None:
                fused_d0b = (_ptr(T['part_ws2']), _ptr(T['amax_ws2']), hb)
        elif hsrc is not None:
            ld0 = self.by_name['d0b']
            L.dc_head_bwd_bnin_bnred(hsrc[0], hsrc[1][0], hsrc[1][1], _ptr(A['p']), yp,
                                     self.pview(self.pflat, lo, 'k'), _ptr(gb[0]), _ptr(T['part_ws']), self.loss_kind,
                                     T['sums'].data_ptr(), self.stat_ptr(ld0, 0), self.stat_ptr(ld0, 1),
                                     _ptr(T['part_ws2']), _ptr(T['amax_ws2']), pixels0, nfb, st)
            fused_d0b = (_ptr(T['part_ws2']), _ptr(T['amax_ws2']), hb)
        else:
            L.dc_head_bwd(_ptr(A['d0b']), _ptr(A['p']), yp, self.pview(self.pflat, lo, 'k'),
                          _ptr(gb[0]), _ptr(T['part_ws']), self.loss_kind, T['sums'].data_ptr(), pixels0, nfb, st)
        L.dc_head_grad_finalize(_ptr(gpart), hb, nfb, self.pview(self.gflat, lo, 'k'),
                                self.pview(self.gflat, lo, 'b'), st)

        main = torch.cuda.current_stream(self.device)
        side = self._side_stream if self.streams == 2 else main
        two = two_streams = side is not main
        mh, sh_ = main.cuda_stream, side.cuda_stream          # raw handles: the hand-offs below go through the C ABI (tape-able)
        if two:
            self._wait_stream(sh_, mh)    # everything queued so far (forward, head) precedes the first wgrad
        S = T['slots']
        slot_free = [None] * S            # event of the weight gradient that last read slot k's dz / table
        g_free = [None] * S               # event of the weight gradient that last read g[k] as `da` (dz on load)
        state = {'slot': 0, 'g': 0}       # next block slot; index of the buffer holding the current activation gradient
        world = parallel.world_size()
        sync = self._sync_bn()

        def g_cur():
            return gb[state['g']]

        def g_next():
            """The buffer the next activation gradient goes to (rotation): safe to overwrite once the weight gradient
            that read it as `da` three blocks ago has finished."""
            k = (state['g'] + 1) % S
            if two and g_free[k] is not None:
                self._wait(mh, g_free[k])
                g_free[k] = None
            return k

        def red_of(la):          # pass-1 sums out of the data gradient's epilogue: dense da of a dropout-free layer only
            return la if la.drop <= 0.0 else None

        def block_bwd(l, x_in, da_ptr, da_ld, dx_ptr, prod=None, fused=None, red=None, da_g=None, on_main=False):
            """da (da_ld-strided) -> gradients of block l; dx (dense [.., cin]) written to dx_ptr unless None.  prod: the
            layer that produced x_in (its activation may be non-materialised: BN + ReLU on load).  fused: (partial, amax,
            rows) when the producer of da emitted the pass-1 sums.  red: the layer whose `da` dx is -- when the data
            gradient runs on the role-split kernel it emits that layer's sums; returned for that layer's `fused`.
            da_g: index of the g buffer da lives in (None: a dcat buffer, written once per step).  on_main: the weight gradient
            stays on the main stream (the step's last block: nothing is left to run beside it)."""
            two = two_streams and not on_main
            bsrc = self._bnin_src(prod, T)
            h, w = self._hw(l.lvl)
            pixels = N * h * w
            z = T['z_' + l.name]
            mptr, keep, seed = self._drop_args(l, masks, step_seed)
            mean, invstd = self.stat_ptr(l, 0), self.stat_ptr(l, 1)
            gamma, beta = self.pview(self.pflat, l, 'gamma'), self.pview(self.pflat, l, 'beta')
            dgamma, dbeta = self.pview(self.gflat, l, 'gamma'), self.pview(self.gflat, l, 'beta')
            dk, ws = self.pview(self.gflat, l, 'k'), _ptr(T['wgrad_ws_main' if on_main else 'wgrad_ws'])
            dz_mode = self._dzin_ok(l, N) if da_ld == l.cout else 0
            if dz_mode == 2 and dx_ptr is None:
                dz_mode = 0
            dzin = dz_mode > 0
            k = state['slot']
            state['slot'] = (k + 1) % S
            if fused is None:
                L.dc_bn_bwd_reduce(da_ptr, da_ld, _ptr(z), mean, invstd, gamma, beta, mptr, keep, seed,
                                   _ptr(T['part_ws']), _ptr(T['amax_ws']) if dzin else None, pixels, l.cout, st)
                fused = (_ptr(T['part_ws']), _ptr(T['amax_ws']), L.dc_bn_bwd_blocks(pixels, l.cout))
            if two_streams and slot_free[k] is not None:   # (also when THIS block's weight gradient stays on the main stream)
                self._wait(mh, slot_free[k])        # the weight gradient that last read this slot has finished
                slot_free[k] = None
            count = float((world if sync else 1) * pixels)
            gg = None
            if sync:
                # (dgamma, dbeta) are adjacent in gflat and stay the rank-LOCAL sums there (the end-of-step all-reduce
                # of gflat makes them global exactly once); dz needs the GLOBAL sums now: all-reduce a scratch copy
                # (a device memcpy, no arithmetic outside the kernels)
                L.dc_bn_bwd_finalize(fused[0], fused[2], l.cout, dgamma, dbeta, st)
                g0, _ = l.off['gamma']
                gg = T['bn_gsum'][:2 * l.cout]
                gg.copy_(self.gflat[g0:g0 + 2 * l.cout])
                self._sync_all_reduce(gg)

            if dzin:
                # ---- dz on load: table -> data gradient (main) -> weight gradient (side); no dz tensor ------------------
                coef = _ptr(T['dz_coef'][k])
                if gg is not None:
                    L.dc_bn_bwd_finalize_dzin(None, fused[1], fused[2], l.cout, mean, invstd, gamma, beta, count,
                                              _ptr(gg), _ptr(gg, l.cout), coef, self.pview(self.gflat, l, 'b'), st)
                else:
                    L.dc_bn_bwd_finalize_dzin(fused[0], fused[1], fused[2], l.cout, mean, invstd, gamma, beta, count,
                                              dgamma, dbeta, coef, self.pview(self.gflat, l, 'b'), st)
                fused_next = None
                xa = (bsrc[0], bsrc[1][0], bsrc[1][1]) if bsrc is not None else (x_in, None, None)
                jrows = L.dc_conv3x3_bwd_joint_blocks(N, h, w, l.cin, l.cout) if (dx_ptr is not None and self.joint) else 0
                if jrows > 0 and l.cin != l.cout and (bsrc is not None or red is not None):
                    jrows = 0          # the 64 -> 32 kernel takes a materialised input and emits no sums
                if jrows > 0:
                    # ---- 32 -> 32 block at a 512^2-class resolution: data AND weight gradient from one kernel, every tensor
                    # of the block read once (csrc/bwd_joint.hip); main stream, nothing left for the side stream
                    rargs = (None,) * 7
                    if red is not None:
                        rargs = (_ptr(T['z_' + red.name]), self.stat_ptr(red, 0), self.stat_ptr(red, 1),
                                 self.pview(self.pflat, red, 'gamma'), self.pview(self.pflat, red, 'beta'),
                                 _ptr(T['part_ws']), _ptr(T['amax_ws']))
                        fused_next = (_ptr(T['part_ws']), _ptr(T['amax_ws']), jrows)
                    L.dc_conv3x3_bwd_joint_f16x3(xa[0], xa[1], xa[2], self._ab_in(l), da_ptr, _ptr(z), coef,
                                                 _ptr(self.wp_dgrad[l.name]), dx_ptr, *rargs, dk, _ptr(T['joint_ws']),
                                                 N, h, w, l.cin, l.cout, st)
                    return fused_next
                if dx_ptr is not None:
                    rargs = (None,) * 7
                    if red is not None:
                        rows = L.dc_conv3x3_dgrad_dzin_blocks(N, h, w, l.cin, l.cout)
                        rargs = (_ptr(T['z_' + red.name]), self.stat_ptr(red, 0), self.stat_ptr(red, 1),
                                 self.pview(self.pflat, red, 'gamma'), self.pview(self.pflat, red, 'beta'),
                                 _ptr(T['part_ws']), _ptr(T['amax_ws']))
                        fused_next = (_ptr(T['part_ws']), _ptr(T['amax_ws']), rows)
                    L.dc_conv3x3_dgrad_dzin_f16x3(da_ptr, _ptr(z), coef, _ptr(self.wp_dgrad[l.name]), dx_ptr,
                                                  _ptr(T['dz'][k]) if dz_mode == 2 else None, *rargs, N, h, w, l.cin, l.cout, st)
                if two:
                    self._wait_stream(sh_, mh)
                sw = st if on_main else sh_
                if dz_mode == 2:
                    # the data gradient has written dz: plain weight gradient, its power-of-two scale from the table's bound row
                    dzp, scale = _ptr(T['dz'][k]), _ptr(T['dz_scale'], 4 * k)
                    L.dc_pow2_scale_from_absmax(coef + 4 * 6 * l.cout, l.cout, 1024.0, scale, sw)
                    if bsrc is not None:
                        L.dc_conv3x3_wgrad_bnin_f16x3(bsrc[0], bsrc[1][0], bsrc[1][1], self._ab_in(l), dzp, dk, ws, scale, N, h, w,
                                                      l.cin, l.cout, sw)
                    else:
                        L.dc_conv3x3_wgrad_f16x3(x_in, dzp, dk, ws, scale, self._ab_in(l), N, h, w, l.cin, l.cout, sw)
                else:
                    L.dc_conv3x3_wgrad_dzin_f16x3(xa[0], xa[1], xa[2], self._ab_in(l), da_ptr, _ptr(z), coef, dk, ws,
                                                  N, h, w, l.cin, l.cout, sw)
                if two:
                    ev = self._record(sh_)
                    slot_free[k] = ev
                    if da_g is not None:
                        g_free[da_g] = ev
                return fused_next

            # ---- apply pass: dz is materialised (Dropout blocks, conv-transposes, shapes the dz-on-load kernels skip) ----
            dz, scale = _ptr(T['dz'][k]), _ptr(T['dz_scale'], 4 * k)
            dpart, amaxp = _ptr(T['dbias_part'][k]), _ptr(T['absmax'][k])
            blocks = L.dc_bn_bwd_blocks(pixels, l.cout)
            if gg is not None:
                L.dc_bn_bwd_apply_count(da_ptr, da_ld, _ptr(z), mean, invstd, gamma, beta, mptr, keep, seed,
                                        _ptr(gg), _ptr(gg, l.cout), dz, dpart, amaxp, pixels, count, l.cout, st)
            else:
                L.dc_bn_bwd_finalize(fused[0], fused[2], l.cout, dgamma, dbeta, st)
                L.dc_bn_bwd_apply(da_ptr, da_ld, _ptr(z), mean, invstd, gamma, beta, mptr, keep, seed, dgamma, dbeta,
                                  dz, dpart, amaxp, pixels, l.cout, st)

            # one launch: conv-bias gradient (column sums of the dz partials) + -- f16x3 -- the exact power-of-two scale
            # that brings max|dz| to [512, 1024] before the fp16 split.  On small steps (launch-latency bound: the
            # reference's own 96^2 / 128^2 training windows) it runs on the weight-gradient stream and the data gradient
            # derives the scale from the per-block maxima itself (+1.3 % there, -0.7 % at 512^2 x 16: DESIGN 5d).
            def finalize(stream):
                L.dc_bn_bwd_apply_finalize(dpart, amaxp if f16 else None, blocks, l.cout, 1024.0,
                                           self.pview(self.gflat, l, 'b'), scale if f16 else None, stream)
            side_fin = f16 and N * self.H * self.W <= (1 << 20)
            if not side_fin:
                finalize(st)
            d_scale, d_amax, d_amax_n = (None, amaxp, blocks) if side_fin else (scale, None, 0)

            def dgrad():
                if l.kind == 'conv':
                    if f16 and red is not None:
                        rows = L.dc_conv3x3_dgrad_bnred_blocks(N, h, w, l.cin, l.cout)
                        if rows > 0:
                            L.dc_conv3x3_dgrad_bnred_f16x3(dz, wpd, dx_ptr, d_scale, d_amax, d_amax_n, _ptr(T['z_' + red.name]),
                                                           self.stat_ptr(red, 0), self.stat_ptr(red, 1),
                                                           self.pview(self.pflat, red, 'gamma'), self.pview(self.pflat, red, 'beta'),
                                                           _ptr(T['part_ws']), _ptr(T['amax_ws']), N, h, w, l.cin, l.cout, st)
                            return (_ptr(T['part_ws']), _ptr(T['amax_ws']), rows)
                    if f16:
                        L.dc_conv3x3_dgrad_f16x3(dz, wpd, dx_ptr, d_scale, d_amax, d_amax_n, _ptr(T['splitk_ws']), N, h, w, l.cin, l.cout, st)
                    else:
                        L.dc_conv3x3_dgrad(dz, wpd, dx_ptr, N, h, w, l.cin, l.cout, st)
                elif f16:
                    rows = L.dc_convT2x2_dgrad_bnred_blocks(N, h // 2, w // 2, l.cin, l.cout) if red is not None else 0
                    if rows > 0:      # dx is `da` of the block in front of the up-convolution: its pass-1 sums from this epilogue
                        L.dc_convT2x2_dgrad_bnred_f16x3(dz, wpd, dx_ptr, d_scale, d_amax, d_amax_n, _ptr(T['z_' + red.name]),
                                                        self.stat_ptr(red, 0), self.stat_ptr(red, 1),
                                                        self.pview(self.pflat, red, 'gamma'), self.pview(self.pflat, red, 'beta'),
                                                        _ptr(T['part_ws']), _ptr(T['amax_ws']), N, h // 2, w // 2, l.cin, l.cout, st)
                        return (_ptr(T['part_ws']), _ptr(T['amax_ws']), rows)
                    L.dc_convT2x2_dgrad_f16x3(dz, wpd, dx_ptr, d_scale, d_amax, d_amax_n, _ptr(T['splitk_ws']), N, h // 2, w // 2, l.cin, l.cout, st)
                else:
                    L.dc_convT2x2_dgrad(dz, wpd, dx_ptr, N, h // 2, w // 2, l.cin, l.cout, st)
                return None

            fused_next = None
            wpd = _ptr(self.wp_dgrad[l.name]) if dx_ptr is not None else None
            # where the data gradient runs on the role-split kernel (it owns a CU's whole LDS: the two cannot share a CU
            # anyway) it goes FIRST and the weight gradient starts when it has finished -- the critical path never queues
            # behind 256 persistent weight-gradient workgroups; elsewhere (conv-transpose, narrow layers) they start together
            after = dx_ptr is not None and f16 and l.kind == 'conv' and L.dc_conv3x3_pp_blocks(N, h, w, l.cin, l.cout, 1, 0) > 0
            if after:
                fused_next = dgrad()
            if two:
                self._wait_stream(sh_, mh)
            sw = st if on_main else sh_
            if side_fin:
                finalize(sw)
            if bsrc is not None and l.kind == 'conv':
                L.dc_conv3x3_wgrad_bnin_f16x3(bsrc[0], bsrc[1][0], bsrc[1][1], self._ab_in(l), dz, dk, ws, scale, N, h, w,
                                              l.cin, l.cout, sw)
            elif bsrc is not None:
                L.dc_convT2x2_wgrad_bnin_f16x3(bsrc[0], bsrc[1][0], bsrc[1][1], self._ab_in(l), dz, dk, ws, scale, N,
                                               h // 2, w // 2, l.cin, l.cout, sw)
            elif l.kind == 'conv':
                if f16:
                    L.dc_conv3x3_wgrad_f16x3(x_in, dz, dk, ws, scale, self._ab_in(l), N, h, w, l.cin, l.cout, sw)
                else:
                    L.dc_conv3x3_wgrad(x_in, dz, dk, ws, N, h, w, l.cin, l.cout, sw)
            elif f16:
                L.dc_convT2x2_wgrad_f16x3(x_in, dz, dk, ws, scale, self._ab_in(l), N, h // 2, w // 2, l.cin, l.cout, sw)
            else:
                L.dc_convT2x2_wgrad(x_in, dz, dk, ws, N, h // 2, w // 2, l.cin, l.cout, sw)
            if two:
                slot_free[k] = self._record(sh_)
            if dx_ptr is not None and not after:
                fused_next = dgrad()
            return fused_next

        # ---- decoder: d<l>b -> d<l>a -> (up-conv | up-sampling) for l = 0..3 ---------------------------------------
        fused_up = None           # pass-1 sums of the block in front of the up-convolution, from that up-convolution's data gradient
        for lvl in (0, 1, 2, 3):
            c = nfb << lvl
            cat, dcat = A['cat%d' % lvl], T['dcat%d' % lvl]
            la = self.by_name['d%da' % lvl]
            ki, ko = state['g'], g_next()
            fa = block_bwd(self.by_name['d%db' % lvl], _ptr(A['d%da' % lvl]), _ptr(gb[ki]), c, _ptr(gb[ko]),
                           prod=la, fused=fused_d0b if lvl == 0 else fused_up, red=red_of(la), da_g=ki)
            state['g'] = ko
            block_bwd(la, _ptr(cat), _ptr(gb[ko]), c, _ptr(dcat), fused=fa, da_g=ko)
            x_up = A['bb'] if lvl == 3 else A['d%db' % (lvl + 1)]
            l_up = self.by_name['bb' if lvl == 3 else 'd%db' % (lvl + 1)]
            kn = g_next()
            if self.upsampling:
                h, w = self._hw(lvl)
                mptr, keep, seed = self._up_drop_args(lvl, masks, step_seed)
                L.dc_upsample2x_drop_bwd(_ptr(dcat), 3 * c, mptr, keep, seed, _ptr(gb[kn]), N, h // 2, w // 2, 2 * c, st)
                fused_up = None
            else:
                fused_up = block_bwd(self.by_name['u%d' % lvl], _ptr(x_up), _ptr(dcat), 2 * c, _ptr(gb[kn]), prod=l_up,
                                     red=red_of(l_up))
            state['g'] = kn

        def bucket_done(i):
            if comm is not None:
                # the collective through the C ABI, recorded like any launch (no cut in the tape), on the WEIGHT-GRADIENT stream: that
                # queue idles ~4 ms of a step, and a stream of its own would be one more HIP stream competing for the 4 hardware queues
                # (measured: with a third stream the side stream landed on the main stream's queue and the step serialised, +1.0 ms).
                # The range is final once BOTH queues have reached this point -- weight gradients here, dbias / dgamma / dbeta / head
                # gradients on the main stream; that event carries the system-scope fence
                if self.ar_buckets > 1:
                    if two:
                        self._wait_stream(sh_, mh, fenced=True)
                    lo, hi = self.grad_buckets()[i]
                    L.dc_comm_all_reduce_sum(comm, _ptr(self.gflat, lo), hi - lo, sh_)
                return
            if bucket_cb is None:
                return
            if two:
                self._wait_stream(sh_, mh, fenced=True)  # dbias / dgamma / dbeta / head gradients are written on the main stream
            L.mark(i)                       # (a tape is cut here: the collective is issued from Python between two segments)
            with torch.cuda.stream(side):
                bucket_cb(*self.grad_buckets()[i])

        bucket_done(0)
        # ---- bottleneck and encoder: (max-pool backward + skip) -> <tag>b -> <tag>a for l = 4..0 --------------------
        for lvl in (4, 3, 2, 1, 0):
            c = nfb << lvl
            h, w = self._hw(lvl)
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            if lvl == 3:
                bucket_done(1)
            fused_pool = fused_up if lvl == 4 else None      # bb: its sums came out of u3's data gradient
            if lvl < 4:
                # g = d(pool output); route through the argmax and add the skip gradient (second half of dcat); the
                # kernel also emits the pooled layer's pass-1 sums while it writes da
                cup = self._cup(lvl)
                lb = self.by_name[tag + 'b']
                ki, ko = state['g'], g_next()
                mptr, keep, seed = self._drop_args(lb, masks, step_seed)
                L.dc_maxpool2x2_bwd_bnred(_ptr(gb[ki]), A['idx%d' % lvl].data_ptr(), _ptr(T['dcat%d' % lvl], cup), cup + c,
                                          _ptr(gb[ko]), _ptr(T['z_' + lb.name]), self.stat_ptr(lb, 0),
                                          self.stat_ptr(lb, 1), self.pview(self.pflat, lb, 'gamma'),
                                          self.pview(self.pflat, lb, 'beta'), mptr, keep, seed,
                                          _ptr(T['part_ws2']), _ptr(T['amax_ws2']), N, h, w, c, st)
                fused_pool = (_ptr(T['part_ws2']), _ptr(T['amax_ws2']), L.dc_maxpool2x2_bwd_blocks(N, h, w, c))
                state['g'] = ko
            la = self.by_name[tag + 'a']
            ki, ko = state['g'], g_next()
            fa = block_bwd(self.by_name[tag + 'b'], _ptr(A[tag + 'a']), _ptr(gb[ki]), c, _ptr(gb[ko]), prod=la,
                           fused=fused_pool, red=red_of(la), da_g=ki)
            state['g'] = ko
            if lvl == 0:
                # the step's last block: its weight gradient stays on the main stream (which has nothing else left), beside the previous
                # block's on the side stream -- one join, one Adam launch (same box: 3.019 -> 2.997 ms at 128^2 x 20, 3.197 -> 3.170 at
                # 96^2 x 32, 17.62 -> 17.58 at 512^2 x 16).  DC_TAIL_MAIN=0: both on the side stream in sequence, Adam + re-pack of
                # everything but the first layer underneath the last one (defer_tail).
                tail_main = two and self.tail_main and bucket_cb is None
                if two and defer_tail and bucket_cb is None and not tail_main:
                    self._tail = self._record(sh_)        # everything the side stream has been given so far
                block_bwd(la, xp, _ptr(gb[ko]), c, None, fused=fa, da_g=ko, on_main=tail_main)
            else:
                kn = g_next()
                block_bwd(la, _ptr(A['pool%d' % (lvl - 1)]), _ptr(gb[ko]), c, _ptr(gb[kn]), fused=fa, da_g=ko)
                state['g'] = kn
        if comm is not None:
            # the last range (the encoder's: 15 % of the bytes, the only exposed part) -- or, with one bucket, everything
            probe = getattr(self, 'ar_probe', None) if not L.recording() else None
            if probe is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
            if two:
                self._wait_stream(sh_, mh, fenced=True)
            lo, hi = self.grad_buckets()[-1] if self.ar_buckets > 1 else (0, self.n_train)
            L.dc_comm_all_reduce_sum(comm, _ptr(self.gflat, lo), hi - lo, sh_)
            if two:
                self._wait_stream(mh, sh_, fenced=True)      # Adam (main stream) reads the reduced gradient
            if probe is not None:
                e1.record(main)
                probe.append((e0, e1))
        elif two and self._tail is None:
            self._wait_stream(mh, sh_, fenced=bucket_cb is not None)    # gflat is complete once both streams have drained

    def _join_side(self):
        side = getattr(self, '_side_stream', None)
        if side is not None:
            self._wait_stream(self._stream(), side.cuda_stream)

    def _settle_tail(self):
        """backward(defer_tail=True) returns with the weight-gradient stream still on the first layer's dW / db (it reads
        x, z, g buffers and writes gflat); only adam_step() joins it underneath its own work.  Anything else that follows --
        another forward_train / backward (an exception between the two calls of a step), grads() -- joins here first."""
        if self._tail is not None:
            self._tail = None
            self._join_side()

    @_on_device
    def adam_step(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-8, grad_scale=1.0):
        """Keras-2.0.6 Adam over the flat buffers; `iterations` counts completed steps."""
        t = self.iterations + 1
        lr_t = float(lr * np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t))
        side = getattr(self, '_side_stream', None)
        # (the tail EVENT, not just its presence: the body bakes `self._wait(st, tail)` in, and which pool event the backward
        # left as its tail depends on the backward variant that ran)
        key = ('adam', self._stream(), side.cuda_stream if side is not None else 0, self._tail, float(beta_1),
               float(beta_2), float(epsilon), float(grad_scale), self._packed_dirty)
        self._taped(key, lambda: self._adam_body(lr_t, beta_1, beta_2, epsilon, grad_scale), lambda: {'lr_t': lr_t},
                    post_attrs=('_packed_dirty', '_tail', '_ev_i'))
        self.iterations = t
        self._fold_dirty = True

    def _adam_body(self, lr_t, beta_1, beta_2, epsilon, grad_scale):
        st = self._stream()
        lr_v = self._v('lr_t', lr_t)

        def adam(lo, hi):
            if hi > lo:
                self.L.dc_adam_step_flat(_ptr(self.pflat, lo), _ptr(self.gflat, lo), _ptr(self.mflat, lo), _ptr(self.vflat, lo),
                                         hi - lo, lr_v, beta_1, beta_2, epsilon, float(grad_scale), st)
        tail, self._tail = self._tail, None
        l0 = self.layers[0]
        lo = l0.off['k'][0]
        hi = l0.off['b'][0] + l0.cout
        if tail is not None and lo % 4 == 0 and hi % 4 == 0 and hi <= self.n_train and l0.name not in self.wp_fwd:
            # backward(defer_tail=True): the weight-gradient stream is still on the first layer's weight / bias gradient
            # (228 us at the benchmark size, with nothing left for the main stream).  Everything else is complete at `tail`:
            # its Adam update and the re-pack of the split-fp16 weight images (the first layer has none) run underneath,
            # the first layer's 320 parameters follow the join.
            self._wait(st, tail)
            adam(0, lo)
            adam(hi, self.n_train)
            self._packed_dirty = True
            self.repack()
            self._wait_stream(st, self._side_stream.cuda_stream)
            adam(lo, hi)
        else:
            if tail is not None:
                self._wait_stream(st, self._side_stream.cuda_stream)
            adam(0, self.n_train)
            self._packed_dirty = True

    def read_sums(self):
        """Host copy of the 12 loss/metric sums of the last forward_train (synchronises the stream)."""
        N = self._last[0]
        return self._train_bufs(N)['sums'].cpu().numpy().copy()

    def grads(self):
        """gflat split per layer: {name: [dk, db, dgamma, dbeta]} (host numpy)."""
        with torch.cuda.device(self.device):
            self._settle_tail()
        g = self.gflat.cpu().numpy()
        out = {}
        for l in self.layers:
            out[l.name] = []
            for key in ('k', 'b', 'gamma', 'beta'):
                if key in l.off:
                    o, shp = l.off[key]
                    out[l.name].append(g[o:o + int(np.prod(shp))].reshape(shp).copy())
        return out

    def batch_stats(self):
        """Last training forward's (mean, invstd) per layer (host numpy)."""
        b = self.bstat.cpu().numpy()
        out = {}
        for l in self.layers:
            if l.kind != 'head':
                o = self._stat_off[l.name]
                out[l.name] = (b[o:o + l.cout].copy(), b[o + l.cout:o + 2 * l.cout].copy())
        return out
