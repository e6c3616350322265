"""Micro-benchmark of the conv-transpose kernels (split-fp16 path) on the four UNet2DS up-layer shapes (batch 16 of 512^2;
`python scripts/one_convT.py small`: batch 20 of the reference's 128^2 training windows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
N = 16
SHAPES = [(256, 64, 32), (128, 128, 64), (64, 256, 128), (32, 512, 256)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    N, SHAPES = 20, [(64, 64, 32), (32, 128, 64), (16, 256, 128), (8, 512, 256)]
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tot = [0, 0, 0]
one = torch.ones(1, device='cuda')
for HW, Ci, Co in SHAPES:
    x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, 2 * HW, 2 * HW, Co, device='cuda')
    K = torch.randn(2, 2, Co, Ci, device='cuda') * 0.05
    wf = torch.empty(L.dc_pack_weights_f16x3_floats(1, Ci, 4 * Co), device='cuda')
    wd = torch.empty(L.dc_pack_weights_f16x3_floats(4, Co, Ci), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wf.data_ptr(), 1, Ci, 4 * Co, 0, 1, Ci, 0, None)
    L.dc_pack_weights_f16x3(K.data_ptr(), wd.data_ptr(), 4, Co, Ci, Co * Ci, Ci, 1, 0, None)
    z = torch.empty(N, 2 * HW, 2 * HW, Co, device='cuda'); dx = torch.empty(N, HW, HW, Ci, device='cuda')
    stats = torch.zeros(L.dc_convT2x2_tiles(N, HW, HW, Co) * 4 * Co * 2 * 16 + 1024, device='cuda')
    dw = torch.empty(4 * Ci * Co, device='cuda'); ws = torch.empty(L.dc_convT2x2_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
    f = lambda: L.dc_convT2x2_fwd_f16x3(x.data_ptr(), wf.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr(), None, None, 0, None, 0, None, 0, N, HW, HW, Ci, Co, None)
    d = lambda: L.dc_convT2x2_dgrad_f16x3(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), one.data_ptr(), None, 0, None, N, HW, HW, Ci, Co, None)
    w = lambda: L.dc_convT2x2_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), one.data_ptr(), None, N, HW, HW, Ci, Co, None)
    tf, td, tw = timeit(f), timeit(d), timeit(w)
    floor = (x.numel() + z.numel()) * 4 / 6.3e9
    tot[0] += tf; tot[1] += td; tot[2] += tw
    print('%4d^2 %3d->%3d | fwd %.3f ms | dgrad %.3f ms | wgrad(+reduce) %.3f ms | hbm floor %.3f ms' % (HW, Ci, Co, tf, td, tw, floor))
print('per step: fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms' % tuple(tot))
