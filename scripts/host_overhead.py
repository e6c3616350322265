"""Host-side cost of enqueuing one train step (no device sync inside the timed region) vs its device time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import unet_hip, Adam
N, HW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 128)
m = unet_hip((HW, HW)); m.compile(Adam(0.002), loss='binary_crossentropy')
x = torch.rand(N, HW, HW, device='cuda'); y = (torch.rand(N, HW, HW, device='cuda') < 0.1).to(torch.uint8)
for _ in range(5): m.train_on_device_batch(x, y)
torch.cuda.synchronize()
e = m.engine
t0 = time.perf_counter()
for _ in range(20):
    e.forward_train(x, y, None); e.backward(); e.adam_step(0.002)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('batch %d of %dx%d: host enqueue %.2f ms/step, total %.2f ms/step' % (N, HW, HW, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    e.forward_train(x, y, None); e.backward(); e.adam_step(0.002)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
