"""Train-step time at the reference's own training configurations (examples: 128x128 windows, batch 20; fit() default:
96x96, batch 32) next to the benchmark configuration: python scripts/small_window_step.py [HW B]..."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd.model import Model, Adam
from oracle import unet_numpy as on
cfgs = [(128, 20), (96, 32), (512, 16)]
if len(sys.argv) > 2:
    cfgs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
dev = torch.device('cuda', 0)
for HW, B in cfgs:
    model = Model((HW, HW), 32, device=dev)
    model.compile(Adam(0.002), 'binary_crossentropy')
    x, y = on.synthetic_batch(B, HW, HW, seed_x=1, seed_y=2)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for _ in range(5): model.train_on_device_batch(xd, yd)
    torch.cuda.synchronize()
    n = 30
    t = time.perf_counter()
    for _ in range(n): model.train_on_device_batch(xd, yd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    px = B * HW * HW
    print('%3d^2 x %2d: %7.3f ms/step  %8.0f windows/s  %6.2f us per 1000 pixels' % (HW, B, dt * 1e3, B / dt, dt * 1e6 / (px / 1e3)))
    del model
