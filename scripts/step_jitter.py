"""Per-step GPU time of N consecutive train steps (one timing event per step end, no host synchronisation inside the run): is the occasional
~100 us gap with both queues empty, seen in rocprofv3 traces of the 128^2 x 20 step, there without the profiler?
    python scripts/step_jitter.py [HW B steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd.model import Model, Adam  # noqa: E402
from oracle import unet_numpy as on  # noqa: E402

HW, B, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 20, 300)
dev = torch.device('cuda', 0)
model = Model((HW, HW), 32, device=dev)
model.compile(Adam(0.002), 'binary_crossentropy')
x, y = on.synthetic_batch(B, HW, HW, seed_x=1, seed_y=2)
xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
for _ in range(10):
    model.train_on_device_batch(xd, yd)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    model.train_on_device_batch(xd, yd)
    ev[i + 1].record()
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)])
print('%d^2 x %d, %d steps: median %.1f us, mean %.1f, p10 %.1f, p90 %.1f, max %.1f; steps > median + 60 us: %d (%.1f %%), their mean excess %.1f us'
      % (HW, B, N, np.median(d), d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max(), (d > np.median(d) + 60).sum(),
         100.0 * (d > np.median(d) + 60).mean(), (d[d > np.median(d) + 60] - np.median(d)).mean() if (d > np.median(d) + 60).any() else 0.0))
print('first 40 (us): ' + ' '.join('%.0f' % v for v in d[:40]))
