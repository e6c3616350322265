import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
HW, Ci, Co, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
dw = torch.empty(3, 3, Ci, Co, device='cuda')
ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
for _ in range(iters):
    L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, None)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters
print('wgrad %d^2 %d->%d: %.3f ms  %.1f TF-equiv' % (HW, Ci, Co, t, 2.0 * 9 * Ci * Co * N * HW * HW / t / 1e9))
