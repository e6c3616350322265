"""Do the weight-gradient kernel and a BatchNorm-backward pass really overlap when put on two streams?
Times wgrad alone, bn_bwd_apply alone, and both launched together (wall time until both are done), per layer shape.
usage: corun.py [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd._lib import lib
L = lib()
N = 16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fns):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(iters):
        for f in fns: f()
    d1, d2 = torch.cuda.Event(), torch.cuda.Event()
    d1.record(s1); d2.record(s2)
    cur.wait_event(d1); cur.wait_event(d2)
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for HW, Ci, Co in [(512, 32, 32), (512, 64, 32), (256, 64, 64), (256, 128, 64), (128, 128, 128), (128, 256, 128), (64, 256, 256), (32, 512, 512)]:
    x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
    dw = torch.empty(3, 3, Ci, Co, device='cuda')
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
    # the BN pass of the NEXT block in backward order works on a tensor of the wgrad's input size
    C = Ci; pixels = N * HW * HW
    da = torch.randn(pixels, C, device='cuda'); z = torch.randn(pixels, C, device='cuda'); dzo = torch.empty_like(z)
    v = [torch.rand(C, device='cuda') + 0.5 for _ in range(6)]
    blocks = L.dc_bn_bwd_blocks(pixels, C)
    p2 = torch.empty(blocks * C, device='cuda'); am = torch.empty(blocks, device='cuda')
    fw = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, s1.cuda_stream)
    fb = lambda: L.dc_bn_bwd_apply(da.data_ptr(), C, z.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                   None, 1.0, 0, v[4].data_ptr(), v[5].data_ptr(), dzo.data_ptr(), p2.data_ptr(), am.data_ptr(), pixels, C, s2.cuda_stream)
    tw, tb, tt = wall([fw]), wall([fb]), wall([fw, fb])
    each = {}
    for order in ('wb', 'bw'):          # one launch each, both dispatch orders: who is starved?
        rec = []
        for _ in range(7):
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record(torch.cuda.current_stream()); s1.wait_event(ev[0]); s2.wait_event(ev[0])
            def one(f, s, a, b):
                ev[a].record(s); f(); ev[b].record(s)
            if order == 'wb': one(fw, s1, 1, 2); one(fb, s2, 3, 4)
            else: one(fb, s2, 3, 4); one(fw, s1, 1, 2)
            torch.cuda.synchronize()
            rec.append((ev[1].elapsed_time(ev[2]) * 1e3, ev[3].elapsed_time(ev[4]) * 1e3, max(ev[0].elapsed_time(ev[2]), ev[0].elapsed_time(ev[4])) * 1e3))
        rec.sort(key=lambda r: r[2])
        each[order] = rec[len(rec) // 2]
    print('%4d^2 %3d->%3d  wgrad %6.1f us  bn_apply(C=%d) %6.1f us  together %6.1f us  (sum %6.1f, max %6.1f)  hidden %.0f %%' % (
        HW, Ci, Co, tw, C, tb, tt, tw + tb, max(tw, tb), 100 * (tw + tb - tt) / min(tw, tb)))
    for order in ('wb', 'bw'):
        print('        launched %s: wgrad %6.1f us, bn_apply %6.1f us, both done after %6.1f us' % ((('wgrad first', 'bn first')[order == 'bw'],) + each[order]))
