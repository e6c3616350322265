"""Kernel micro-benchmark: conv3x3 fwd fp32-MFMA vs split-fp16 (f16x3) on the UNet2DS layer shapes (batch 16)."""
import ctypes, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(512, 32, 32), (512, 64, 32), (256, 32, 64), (256, 64, 64), (256, 128, 64), (128, 64, 128), (128, 128, 128),
          (128, 256, 128), (64, 128, 256), (64, 256, 256), (64, 512, 256), (32, 256, 512), (32, 512, 512)]
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tot32 = tot16 = 0
for HW, Ci, Co in shapes:
    rs = np.random.RandomState(0)
    x = torch.randn(N, HW, HW, Ci, device='cuda')
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    Kd = torch.from_numpy(K).cuda()
    wp = torch.empty(9 * Ci * Co, device='cuda'); wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    L.dc_pack_weights(Kd.data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    z32 = torch.empty(N, HW, HW, Co, device='cuda'); z16 = torch.empty_like(z32)
    f32 = lambda: L.dc_conv3x3_fwd(x.data_ptr(), wp.data_ptr(), None, z32.data_ptr(), Co, None, None, None, 0, N, HW, HW, Ci, Co, None)
    f16 = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z16.data_ptr(), Co, None, 0, None, None, 0, None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
    t32, t16 = timeit(f32), timeit(f16)
    fl = 2.0 * 9 * Ci * Co * N * HW * HW
    # accuracy on image 0 vs float64
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), torch.from_numpy(K).permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1).numpy()
    e32 = np.abs(z32[:1].cpu().numpy() - ref).max() / np.abs(ref).max(); e16 = np.abs(z16[:1].cpu().numpy() - ref).max() / np.abs(ref).max()
    tot32 += t32; tot16 += t16
    print('%4d^2 %3d->%3d  fp32 %.3f ms %6.1f TF  | f16x3 %.3f ms %6.1f TF-equiv  speedup %.2fx | relerr fp32 %.1e f16x3 %.1e' % (HW, Ci, Co, t32, fl / t32 / 1e9, t16, fl / t16 / 1e9, t32 / t16, e32, e16))
print('total fp32 %.2f ms  f16x3 %.2f ms' % (tot32, tot16))
