"""Which shared resource slows the weight-gradient kernel when another kernel runs beside it?  wgrad (256^2, 64->64) is
launched first, then ONE synthetic co-runner (scripts/micro/corunners.hip) sized to last about as long: pure VALU,
HBM reads, HBM reads + writes, each with 256 / 1024 / 2048 workgroups of 256 threads."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
M = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libcorunners.so'))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
M.valu_spin.argtypes = [vp, i32, i32, vp]; M.stream_read.argtypes = [vp, i64, vp, i32, vp]; M.stream_copy.argtypes = [vp, vp, i64, i32, vp]
N, HW, Ci, Co = 16, 256, 64, 64
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
dw = torch.empty(3, 3, Ci, Co, device='cuda'); ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
src = torch.randn(N * HW * HW * 64 * 2, device='cuda'); dst = torch.empty_like(src); out = torch.zeros(4, device='cuda')
n16 = src.numel() // 4
fw = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, s1.cuda_stream)


def pair(fo):
    rec = []
    for _ in range(7):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(torch.cuda.current_stream()); s1.wait_event(ev[0]); s2.wait_event(ev[0])
        ev[1].record(s1); fw(); ev[2].record(s1)
        if fo is not None:
            ev[3].record(s2); fo(); ev[4].record(s2)
        torch.cuda.synchronize()
        rec.append((ev[1].elapsed_time(ev[2]) * 1e3, ev[3].elapsed_time(ev[4]) * 1e3 if fo is not None else 0.0))
    rec.sort()
    return rec[len(rec) // 2]


def alone(fo):
    fo(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s2); fo(); e1.record(s2); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


fw(); torch.cuda.synchronize()
print('wgrad alone: %.1f us' % pair(None)[0])
for blocks in (256, 1024, 2048):
    for name, fo in [('valu spin', lambda: M.valu_spin(out.data_ptr(), blocks, 6000 * 256 // blocks * 4, s2.cuda_stream)),
                     ('hbm read', lambda: M.stream_read(src.data_ptr(), n16, out.data_ptr(), blocks, s2.cuda_stream)),
                     ('hbm copy', lambda: M.stream_copy(src.data_ptr(), dst.data_ptr(), n16 // 2, blocks, s2.cuda_stream))]:
        ta = alone(fo)
        tw, to = pair(fo)
        print('%5d blocks  %-10s alone %6.1f us | beside wgrad: wgrad %6.1f us, co-runner %6.1f us' % (blocks, name, ta, tw, to))
