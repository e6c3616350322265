"""Host-side input pipeline of the data-parallel fit() ("must not starve 8 GPUs at global batch 128"):
time per batch of the RNG-exact generator at 512x512 -- the whole global batch (what every rank did in round 1) vs this
rank's slice only (all draws made, 1/8 of the array work) -- against the 20.4 ms the GPU step takes.  CPU only.
    python scripts/input_pipeline_speed.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd.unet2ds import UNet2DSummary, _flatten_mask_stack      # noqa: E402

rs = np.random.RandomState(0)
S, M = [], []
for k in range(19):
    S.append(rs.standard_normal((512, 512)).astype(np.float32))
    stack = np.zeros((120, 512, 512), np.int8)
    for z in range(120):
        cy, cx = rs.randint(8, 504, 2)
        stack[z, cy - 4:cy + 5, cx - 4:cx + 5] = 1
    t0 = time.perf_counter()
    M.append(_flatten_mask_stack(stack))
    tm = time.perf_counter() - t0
print('_summarize_mask (vectorised), 120 neurons on 512x512: %.1f ms per dataset' % (tm * 1e3))
m = UNet2DSummary.__new__(UNet2DSummary)
names = ['d%d' % k for k in range(19)]
yc = [(0, 384)] * 19
for label, shard in (('global batch 128 (every rank, round 1)', None), ('slice of 16 = rank 0 of 8 (round 2)', (0, 8))):
    np.random.seed(865)
    gen = m._batch_gen(S, M, names, yc, 128, 100, (512, 512), 15, shard=shard)
    next(gen)
    ts = []
    for _ in range(12 if shard is None else 40):
        t0 = time.perf_counter()
        sb, mb = next(gen)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print('%-42s median %6.1f ms (min %5.1f) per batch -> %5.1f batches/s (GPU step: 20.4 ms = 49 steps/s)  slice %r'
          % (label, dt * 1e3, min(ts) * 1e3, 1 / dt, sb.shape))
