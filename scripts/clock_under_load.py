"""What shader clock does the chip hold while (a) nothing, (b) the weight-gradient kernel, (c) a BatchNorm-backward pass,
(d) both, (e) the role-split conv kernel run?  A probe wave (scripts/micro/clock_probe.hip) compares the shader cycle
counter with the constant 100 MHz counter while the load loops on other streams."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libclock_probe.so'))
P.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]
N, HW, Ci, Co = 16, 256, 64, 64
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
dw = torch.empty(3, 3, Ci, Co, device='cuda'); ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
C = Ci; pixels = N * HW * HW
da = torch.randn(pixels, C, device='cuda'); z = torch.randn(pixels, C, device='cuda'); dzo = torch.empty_like(z)
v = [torch.rand(C, device='cuda') + 0.5 for _ in range(6)]
blocks = L.dc_bn_bwd_blocks(pixels, C)
p2 = torch.empty(blocks * C, device='cuda'); am = torch.empty(blocks, device='cuda')
K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
zc = torch.empty(N, HW, HW, Co, device='cuda')
fw = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, s1.cuda_stream)
fb = lambda: L.dc_bn_bwd_apply(da.data_ptr(), C, z.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                               None, 1.0, 0, v[4].data_ptr(), v[5].data_ptr(), dzo.data_ptr(), p2.data_ptr(), am.data_ptr(), pixels, C, s2.cuda_stream)
fc = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, zc.data_ptr(), Co, None, 0, None, None, 0, None, 0, None, 0, None, N, HW, HW, Ci, Co, s1.cuda_stream)
out = torch.zeros(2, dtype=torch.int64, device='cuda')
for name, fns in [('idle', []), ('wgrad', [fw]), ('bn_apply', [fb]), ('wgrad + bn_apply', [fw, fb]), ('conv fwd (pp)', [fc]), ('conv fwd + bn_apply', [fc, fb])]:
    for rep in range(2):
        torch.cuda.synchronize()
        for _ in range(60):                      # ~15-20 ms of load queued
            for f in fns: f()
        for _ in range(3000): pass
        P.clock_probe(out.data_ptr(), 500000, s3.cuda_stream)   # 5 ms window in the middle of it
        torch.cuda.synchronize()
    r, c = out.tolist()
    print('%-22s shader clock %.0f MHz over %.2f ms' % (name, c / r * 100.0, r / 1e5))
