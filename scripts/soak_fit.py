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

    def on_epoch_end(self, epoch, logs=None):
        torch.cuda.synchronize()
        now = time.time()
        rss_now = int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 30     # CURRENT resident set
        print('epoch %d: %.1f steps/s incl. validation (%.2f ms/step: steps %.1f ms + epoch end %.1f ms), loss %.4f F1 %.3f val_nf_f1 %.3f | device %.2f GB allocated, %.2f GB reserved, host RSS now %.3f GB (peak %.2f GB)'
              % (epoch, steps / (now - self.t), (now - self.t) / steps * 1e3, (self.t_steps - self.t) * 1e3, (now - self.t_steps) * 1e3,
                 logs['loss'], logs['F1'], logs.get('val_nf_f1_mean', float('nan')),
                 torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30, rss_now,
                 resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20), flush=True)


np.random.seed(865)
model = UNet2DSummary(cpdir=os.path.join(tmp, 'cp'))
hist, _ = model.fit(paths, shape_trn=(window, window), shape_val=(hw, hw), batch_size_trn=batch, nb_steps_trn=steps, nb_epochs=epochs,
                    keras_callbacks=[Probe()], prop_trn=0.75, prop_val=0.25)
print('loss per epoch', ['%.4f' % v for v in hist['loss']])
assert hist['loss'][-1] < hist['loss'][0]
