"""Where does a launch of the two persistent matrix kernels spend its time OUTSIDE the steady-state loop?  Every workgroup stamps
the chip-wide 100 MHz counter (s_memrealtime) at kernel entry, when its first tile / stage is ready, when its loop is done and
when its last store has left (builds with -DDC_WG_CLOCK / -DDC_PP_TIMELINE); the script lines the 256 workgroups up on one time
axis.  Columns (us): launch span = last exit - first entry; entry skew = last entry - first entry; prologue / loop / tail = medians
over workgroups (p10-p90 in brackets); exit skew = last exit - median exit; event = HIP-event time per launch back to back.
    python scripts/kernel_timeline.py build      # here
    python scripts/kernel_timeline.py run        # on the GPU box"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_timeline.so')
SHAPES = [(128, 128, 128), (64, 256, 256), (256, 64, 64), (32, 512, 512)]


def report(tag, tl, n_wg, us_event):
    t = tl[:n_wg * 4].reshape(n_wg, 4).astype(np.float64) * 0.01      # us
    t0 = t[:, 0].min()
    pro, loop, tail = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    q = lambda v: '%6.1f [%5.1f-%5.1f]' % (np.median(v), np.percentile(v, 10), np.percentile(v, 90))
    print('%-34s span %6.1f | entry skew %5.1f | prologue %s | loop %s | tail %s | exit skew %5.1f | event %6.1f'
          % (tag, t[:, 3].max() - t0, t[:, 0].max() - t0, q(pro), q(loop), q(tail), t[:, 3].max() - np.median(t[:, 3]), us_event), flush=True)


def timed(run, torch, iters=20):
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def worker():
    import torch
    from deep_calcium_amd._lib import lib
    from scripts.igemm_pp_ablate import make_run
    L = lib()
    f_wg, f_pp = L.cdll.dc_debug_wgrad_timeline, L.cdll.dc_debug_pp_timeline
    f_wg.argtypes = f_pp.argtypes = [ctypes.c_void_p]
    buf = np.zeros(4096, np.uint64)
    N = 16
    for HW, Ci, Co in SHAPES:
        x = torch.randn(N, HW, HW, Ci, device='cuda'); dz = torch.randn(N, HW, HW, Co, device='cuda')
        dw = torch.empty(3, 3, Ci, Co, device='cuda')
        ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
        run = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, None)
        us = timed(run, torch)
        f_wg(buf.ctypes.data)
        report('wgrad %d^2 %d->%d (+2 reduce launches)' % (HW, Ci, Co), buf.copy(), 256, us)
        del x, dz, dw, ws
    for variant in (0, 3, 1):
        for HW, Ci, Co in SHAPES:
            run, keep = make_run(L, variant, HW, Ci, Co, N)
            us = timed(run, torch)
            f_pp(buf.ctypes.data)
            report('igemm_pp<2,2,%d> %d^2 %d->%d' % (variant, HW, Ci, Co), buf.copy(), 256, us)
            del run, keep


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        from scripts.build_variant import build_variant
        print(build_variant('timeline', ['wgrad_f16x3.hip', 'igemm_pp.hip'], ['-DDC_WG_CLOCK', '-DDC_PP_TIMELINE']))
    elif sys.argv[1] == '--worker':
        worker()
    else:
        subprocess.run([sys.executable, os.path.abspath(__file__), '--worker'], check=False, env=dict(os.environ, DC_LIB_PATH=LIB))
