"""Per-layer cost table of the split-fp16 path (single stream): conv3x3 fwd (with BN partials), dgrad, wgrad on the
UNet2DS layer shapes at batch N; TF-equivalent rates and the HBM floor of each launch."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(512, 32, 32), (512, 64, 32), (256, 32, 64), (256, 64, 64), (256, 128, 64), (128, 64, 128), (128, 128, 128),
          (128, 256, 128), (64, 128, 256), (64, 256, 256), (64, 512, 256), (32, 256, 512), (32, 512, 512)]
count = {(512, 32, 32): 2, (512, 64, 32): 1, (256, 32, 64): 1, (256, 64, 64): 2, (256, 128, 64): 1, (128, 64, 128): 1,
         (128, 128, 128): 2, (128, 256, 128): 1, (64, 128, 256): 1, (64, 256, 256): 2, (64, 512, 256): 1,
         (32, 256, 512): 1, (32, 512, 512): 1}
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tot = [0, 0, 0]
one = torch.ones(1, device='cuda')
for HW, Ci, Co in shapes:
    x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
    K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
    wf = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    wd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Co, Ci), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wf.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    L.dc_pack_weights_f16x3(K.data_ptr(), wd.data_ptr(), 9, Co, Ci, Ci * Co, 1, Co, 1, None)
    z = torch.empty(N, HW, HW, Co, device='cuda'); dx = torch.empty(N, HW, HW, Ci, device='cuda')
    stats = torch.zeros(L.dc_conv3x3_tiles(N, HW, HW, Co) * 2 * Co + 1024, device='cuda', dtype=torch.float64)
    dw = torch.empty(9 * Ci * Co, device='cuda'); ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
    f = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wf.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr() if stats is not None else None, 0, None, None, 0, None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
    d = lambda: L.dc_conv3x3_dgrad_f16x3(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), one.data_ptr(), None, 0, None, N, HW, HW, Ci, Co, None)
    w = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), one.data_ptr(), None, N, HW, HW, Ci, Co, None)
    tf, td, tw = timeit(f), timeit(d), timeit(w)
    fl = 2.0 * 9 * Ci * Co * N * HW * HW
    floor = (Ci + Co) * 4.0 * N * HW * HW / 6.3e9   # ms at 6.3 TB/s
    c = count[(HW, Ci, Co)]
    tot[0] += c * tf; tot[1] += c * td; tot[2] += c * tw
    print('%4d^2 %3d->%3d x%d | fwd %.3f ms %5.0f TF | dgrad %.3f ms %5.0f TF | wgrad(+reduce) %.3f ms %5.0f TF | hbm floor %.3f ms'
          % (HW, Ci, Co, c, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9, floor))
print('per step (22 layers): fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms' % tuple(tot))
