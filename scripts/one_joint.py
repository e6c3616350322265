"""Stand-alone timing of dc_conv3x3_bwd_joint_f16x3 at the benchmark shape (batch 16 of 512^2 x 32 -> 32) against the two
separate dz-on-load kernels:  python scripts/one_joint.py"""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
N, H, W, C = 16, 512, 512, 32
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(N, H, W, C, device='cuda', generator=g); z = torch.randn(N, H, W, C, device='cuda', generator=g)
da = torch.randn(N, H, W, C, device='cuda', generator=g) * 1e-3
K = torch.randn(3, 3, C, C, device='cuda', generator=g) * 0.05
mean = z.view(-1, C).mean(0).contiguous(); invstd = (1 / torch.sqrt(z.view(-1, C).var(0, unbiased=False) + 1e-3)).contiguous()
gamma = torch.ones(C, device='cuda'); beta = torch.zeros(C, device='cuda')
xsc = torch.ones(C, device='cuda'); xsh = torch.zeros(C, device='cuda')
M = N * H * W
blocks = L.dc_bn_bwd_blocks(M, C)
part = torch.empty(blocks * C * 2, device='cuda'); amx = torch.empty(blocks * C, device='cuda')
L.dc_bn_bwd_reduce(da.data_ptr(), C, z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 1.0, 0, part.data_ptr(), amx.data_ptr(), M, C, None)
dg, db, dbias = (torch.empty(C, device='cuda') for _ in range(3))
coef = torch.empty(7 * C, device='cuda')
L.dc_bn_bwd_finalize_dzin(part.data_ptr(), amx.data_ptr(), blocks, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(M), dg.data_ptr(), db.data_ptr(), coef.data_ptr(), dbias.data_ptr(), None)
wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, C, C), device='cuda')
L.dc_pack_weights_f16x3(K.data_ptr(), wpd.data_ptr(), 9, C, C, C * C, 1, C, 1, None)
dx = torch.empty_like(x); dw = torch.empty(3, 3, C, C, device='cuda')
rows = L.dc_conv3x3_bwd_joint_blocks(N, H, W, C, C)
p2 = torch.empty(max(rows, L.dc_conv3x3_dgrad_dzin_blocks(N, H, W, C, C)) * C * 2, device='cuda'); a2 = torch.empty_like(p2)
ws = torch.empty(max(L.dc_conv3x3_bwd_joint_ws_floats(N, H, W, C, C), L.dc_conv3x3_wgrad_ws_floats(N, H, W, C, C)), device='cuda')
red = (x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), p2.data_ptr(), a2.data_ptr())
def joint(r):
    L.dc_conv3x3_bwd_joint_f16x3(x.data_ptr(), xsc.data_ptr(), xsh.data_ptr(), None, da.data_ptr(), z.data_ptr(), coef.data_ptr(), wpd.data_ptr(),
                                 dx.data_ptr(), *(red if r else (None,) * 7), dw.data_ptr(), ws.data_ptr(), N, H, W, C, C, None)
def sep(r):
    L.dc_conv3x3_dgrad_dzin_f16x3(da.data_ptr(), z.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, *(red if r else (None,) * 7), N, H, W, C, C, None)
    L.dc_conv3x3_wgrad_dzin_f16x3(x.data_ptr(), xsc.data_ptr(), xsh.data_ptr(), None, da.data_ptr(), z.data_ptr(), coef.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, C, C, None)
for name, fn in (('joint + sums', lambda: joint(True)), ('joint', lambda: joint(False)), ('separate + sums', lambda: sep(True))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print('%-18s %.1f us' % (name, e0.elapsed_time(e1) * 100))
