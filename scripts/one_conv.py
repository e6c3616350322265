import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
HW, Ci, Co, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(N, HW, HW, Ci, device='cuda')
K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
z = torch.empty(N, HW, HW, Co, device='cuda')
for _ in range(iters):
    L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, None, 0, None, None, 0, None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
torch.cuda.synchronize()
