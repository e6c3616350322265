"""Soak: the example's own training configuration (128x128 windows, batch 20, 100 steps per epoch, validation at 512x512 every
epoch) for a few epochs on a synthetic Neurofinder directory; reports steps/s per epoch, device / host memory growth and the loss.
    python scripts/soak_fit.py [epochs=3] [steps=100] [window=128] [batch=20] [datasets=2]
window 512, batch 16: the benchmark configuration through fit() -- the dz-on-load / joint backward kernels inside real training.
datasets 19: the reference example's own run (all 19 Neurofinder training sets: 114 validation forwards + scorings per epoch).
Prints, per epoch, the wall time split into training steps and the epoch-end callbacks (validation, CSV, checkpoint)."""
import os, resource, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from _nf_dirs import make_neurofinder_dir
from deep_calcium_amd import UNet2DSummary
from deep_calcium_amd.nf_datasets import nf_load_hdf5
from deep_calcium_amd.model import Callback

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
window = int(sys.argv[3]) if len(sys.argv) > 3 else 128
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 20
hw = max(512, (window * 3 // 2 + 15) // 16 * 16)           # the training crop comes from the upper 75 % of the image
nds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
tmp = tempfile.mkdtemp(prefix='dc_soak_')
from deep_calcium_amd.nf_datasets import NEUROFINDER_NAMES
ds_names = [n for n in NEUROFINDER_NAMES if '.test' not in n][:nds]
for k, name in enumerate(ds_names):
    make_neurofinder_dir(tmp, name, hw=(hw, hw), seed=5 + k, frames=3)
paths = nf_load_hdf5(','.join(ds_names), datasets_dir=tmp)


class Probe(Callback):
    """Last in the callback list: on_epoch_end runs after validation / CSV / checkpoint / LR plateau; on_batch_end of the
    epoch's last step marks where the training steps end (host time: the step's metrics are back)."""
    def on_epoch_begin(self, epoch, logs=None):
        self.t = time.time()

    def on_batch_end(self, batch, logs=None):
        if batch == steps - 1:
            self.t_steps = time.time()
        if os.environ.get('SOAK_DETAIL') == 'steps' and batch in (0, 1, 2, 3, 5, 10, 50, steps - 1):
            print('   step %d: RSS %.1f MB' % (batch, int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20), flush=True)

    def on_epoch_end(self, epoch, logs=None):
        torch.cuda.synchronize()
        now = time.time()
        rss_now = int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 30     # CURRENT resident set
        if os.environ.get('SOAK_DETAIL'):
            import tracemalloc, gc
            if epoch == 2:
                tracemalloc.start(12)
                self.snap0 = None
            if epoch == 3:
                gc.collect(); self.snap0 = tracemalloc.take_snapshot()
            if epoch == 6 and getattr(self, 'snap0', None) is not None:
                gc.collect()
                top = tracemalloc.take_snapshot().compare_to(self.snap0, 'traceback')
                for t in top[:6]:
                    print('   tracemalloc +%.1f MB in %d blocks' % (t.size_diff / 2 ** 20, t.count_diff))
                    for line in t.traceback.format()[-8:]:
                        print('      ' + line.strip())
                tracemalloc.stop()
            roll = dict((l.split(':')[0], int(l.split()[1]) / 2 ** 20) for l in open('/proc/self/smaps_rollup').read().splitlines()[1:] if l.split()[1].isdigit())
            hs = {}
            try:
                hs = torch.cuda.host_memory_stats()
            except Exception as e:
                hs = {'err': str(e)[:40]}
            print('   smaps GB: anon %.3f file %.3f shmem %.3f | pinned allocator: %s | threads %d | /dev/shm %s' % (
                roll.get('Anonymous', 0), roll.get('Pss_File', 0), roll.get('Pss_Shmem', 0),
                {k: round(v / 2 ** 30, 3) for k, v in hs.items() if 'bytes' in k and ('current' in k or 'peak' in k) and isinstance(v, (int, float))},
                len(os.listdir('/proc/self/task')), [f for f in os.listdir('/dev/shm')][:4]), flush=True)
        print('epoch %d: %.1f steps/s incl. validation (%.2f ms/step: steps %.1f ms + epoch end %.1f ms), loss %.4f F1 %.3f val_nf_f1 %.3f | device %.2f GB allocated, %.2f GB reserved, host RSS now %.3f GB (peak %.2f GB)'
              % (epoch, steps / (now - self.t), (now - self.t) / steps * 1e3, (self.t_steps - self.t) * 1e3, (now - self.t_steps) * 1e3,
                 logs['loss'], logs['F1'], logs.get('val_nf_f1_mean', float('nan')),
                 torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30, rss_now,
                 resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20), flush=True)


abl = os.environ.get('SOAK_ABL', '')           # leak hunting: switch one epoch-end activity off
if abl:
    import deep_calcium_amd.model as _m
    import deep_calcium_amd.unet2ds as _u
    if abl == 'nockpt':
        _m.ModelCheckpoint.on_epoch_end = lambda self, epoch, logs=None: None
    elif abl == 'blocking':
        _orig = _m.Model.save
        _m.Model.save = lambda self, path, include_optimizer=True, background=False: _orig(self, path, include_optimizer, False)
    elif abl == 'noval':
        def _noval(self, epoch, logs={}):
            for k in ('val_nf_f1_mean', 'val_nf_f1_median', 'val_nf_f1_min', 'val_nf_f1_adj', 'val_nf_prec', 'val_nf_reca'):
                logs[k] = 0.5
        _u._ValidationMetricsCB.on_epoch_end = _noval
    elif abl == 'val_noscore':
        import deep_calcium_amd.nf_metrics as _n
        _n.NativeScorer.__call__ = lambda self, m, mp: (0.5, 0.5, 0., 0., 0.5)
    elif abl == 'val_noweights':     # no weight copies into the validation engines, no re-pack / re-fold
        _m.Model.copy_weights_from = lambda self, other: None
        import torch as _t
        _orig_copy = _t.Tensor.copy_
    elif abl == 'val_nofwd':
        import deep_calcium_amd.net as _ne
        _cache = {}
        _of = _ne.UNetEngine.forward_infer
        def _fi(self, x):
            k = (id(self), x.shape[0])
            if k not in _cache:
                _cache[k] = _of(self, x)
            return _cache[k]
        _ne.UNetEngine.forward_infer = _fi
    elif abl == 'val_probe':
        _orig_oee = _u._ValidationMetricsCB.on_epoch_end
        def _rss():
            return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20
        def _probe(self, epoch, logs={}):
            _orig_oee(self, epoch, logs)
            if epoch != 1:
                return
            import gc
            n = len(self.S_summ)
            def rep(name, fn, k=5):
                torch.cuda.synchronize(); r0 = _rss()
                for _ in range(k): fn()
                torch.cuda.synchronize(); print('   probe %-40s %+7.1f MB per call' % (name, (_rss() - r0) / k), flush=True)
            rep('_score_on_device', lambda: self._score_on_device(n))
            rep('copy_weights_from', lambda: self.model_val.copy_weights_from(self.model))
            rep('whole on_epoch_end', lambda: _orig_oee(self, 5, {}))
            st = self._dev
            print('   groups', [(k, len(g['ks']), g['eng'] is self.model_val.engine) for k, g in st['groups'].items()], flush=True)
            for kk, g in st['groups'].items():
                ge = g['eng']
                rep('forward_infer batch 8 of %r' % (kk,), lambda: ge.forward_infer(g['x'][0:8]))
                rep('forward_infer batch %d of %r' % (min(len(g['ks']) % 8 or 8, 8), kk), lambda: ge.forward_infer(g['x'][0:(len(g['ks']) % 8 or 8)]))
                def dirty():
                    ge._packed_dirty = ge._fold_dirty = True
                    ge.forward_infer(g['x'][0:8])
                rep('dirty + forward_infer of %r' % (kk,), dirty)
                print('   tapes', len(ge._tapes), 'epoch', ge._tape_epoch, 'bufs', [k for k in ge._bufs if not isinstance(k, str)][:6], flush=True)
        _u._ValidationMetricsCB.on_epoch_end = _probe
    elif abl == 'val_trace':
        def _rss():
            return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20
        def wrap(cls, name):
            orig = getattr(cls, name)
            def w(self, *a, **k):
                r0 = _rss(); out = orig(self, *a, **k); torch.cuda.synchronize()
                print('   %-28s %+7.1f MB' % (name, _rss() - r0), flush=True)
                return out
            setattr(cls, name, w)
        wrap(_m.Model, 'copy_weights_from'); wrap(_u._ValidationMetricsCB, '_forward_chunks'); wrap(_u._ValidationMetricsCB, '_score_on_device')
        wrap(_u._ValidationMetricsCB, 'on_epoch_end'); wrap(_m.CSVLogger, 'on_epoch_end'); wrap(_m.ModelCheckpoint, 'on_epoch_end')
        import deep_calcium_amd.net as _ne
        wrap(_ne.UNetEngine, 'forward_infer'); wrap(_ne.UNetEngine, 'repack'); wrap(_ne.UNetEngine, 'refold')
    elif abl in ('sync_before_val', 'sync_after_val', 'sync_after_fwd', 'sync_every_chunk'):
        _oo = _u._ValidationMetricsCB.on_epoch_end
        def _w(self, epoch, logs={}):
            if abl == 'sync_before_val': torch.cuda.synchronize()
            _oo(self, epoch, logs)
            if abl == 'sync_after_val': torch.cuda.synchronize()
        _u._ValidationMetricsCB.on_epoch_end = _w
        if abl in ('sync_after_fwd', 'sync_every_chunk'):
            import deep_calcium_amd.net as _ne
            _of = _ne.UNetEngine.forward_infer
            cnt = [0]
            def _fi(self, x):
                out = _of(self, x)
                cnt[0] += 1
                if abl == 'sync_every_chunk' or cnt[0] % 4 == 0:
                    torch.cuda.synchronize()
                return out
            _ne.UNetEngine.forward_infer = _fi
    elif abl == 'nofeed':          # snapshot to pinned memory, but no raw file / writer process
        def _save(self, path, include_optimizer=True, background=False):
            self.wait_for_saves()
            self._snapshot(include_optimizer and self.optimizer is not None, sync=True)
        _m.Model.save = _save
np.random.seed(865)
model = UNet2DSummary(cpdir=os.path.join(tmp, 'cp'))
hist, _ = model.fit(paths, shape_trn=(window, window), shape_val=(hw, hw), batch_size_trn=batch, nb_steps_trn=steps, nb_epochs=epochs,
                    keras_callbacks=[Probe()], prop_trn=0.75, prop_val=0.25)
print('loss per epoch', ['%.4f' % v for v in hist['loss']])
assert hist['loss'][-1] < hist['loss'][0]
