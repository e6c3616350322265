"""A/B and ablation of the split-fp16 weight-gradient kernel (csrc/wgrad_f16x3.hip): builds copies of the library under the
kernel's experiment switches (HERE, with hipcc: they travel to the GPU box) and times each on the two shapes that carry
most of the step's weight-gradient time, batch 16.  Every build carries -DDC_WG_CLOCK: the clock column is the shader clock
workgroup 0 saw inside its tile loop (s_memtime / s_memrealtime), busy = MFMA issue cycles / loop cycles of that wave.
    python scripts/wgrad_variants.py build        # here
    python scripts/wgrad_variants.py run [tags]   # on the GPU box
"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ('base', []),
    ('xtile0', ['-DDC_WG_XTILE=0']),
    ('abl_nostage', ['-DDC_WG_ABL=1']),
    ('abl_noload', ['-DDC_WG_ABL=2']),
    ('abl_consumers_alone', ['-DDC_WG_ABL=3']),
    ('abl_producers_alone', ['-DDC_WG_ABL=4']),
    ('abl_loads_alone', ['-DDC_WG_ABL=5']),
    ('abl_noslab', ['-DDC_WG_ABL=8']),
    ('abl_mfma_only', ['-DDC_WG_ABL=11']),
    ('abl_presplit', ['-DDC_WG_ABL=16']),
    ('prio1', ['-DDC_WG_PRIO=1']),
    ('prio3', ['-DDC_WG_PRIO=3']),
    ('depth2', ['-DDC_WG_DEPTH=2']),
    ('depth3', ['-DDC_WG_DEPTH=3']),
    ('rw8', ['-DDC_WG_RW=8']),
    ('rw8_depth2', ['-DDC_WG_RW=8', '-DDC_WG_DEPTH=2']),
    ('rw8_depth2_prio1', ['-DDC_WG_RW=8', '-DDC_WG_DEPTH=2', '-DDC_WG_PRIO=1']),
]
SHAPES = [(128, 128, 128), (64, 256, 256), (256, 64, 64), (32, 512, 512)]


def lib_of(tag):
    return os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_wg_%s.so' % tag)


def worker(tag):
    import torch
    from deep_calcium_amd._lib import lib
    L = lib()
    fn = L.cdll.dc_debug_wgrad_stamps
    fn.argtypes = [ctypes.c_void_p]
    N = 16
    out = []
    for HW, Ci, Co in SHAPES:
        g = torch.Generator(device='cuda'); g.manual_seed(HW * 7 + Ci)
        x = torch.randn(N, HW, HW, Ci, device='cuda', generator=g)
        dz = torch.randn(N, HW, HW, Co, device='cuda', generator=g)
        if tag == 'abl_presplit':       # the same VALUES, stored pre-split: the kernel's MFMA operands (and dW) are those of the base run
            from scripts._presplit import presplit_pack
            x, dz = presplit_pack(x), presplit_pack(dz)
        dw = torch.empty(3, 3, Ci, Co, device='cuda')
        ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, HW, HW, Ci, Co), device='cuda')
        run = lambda: L.dc_conv3x3_wgrad_f16x3(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), None, None, N, HW, HW, Ci, Co, None)
        for _ in range(10): run()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100.0)
        us = float(np.median(ts))
        st = (ctypes.c_ulonglong * 4)()
        fn(ctypes.addressof(st))
        cyc, rt, nt, groups = [int(v) for v in st]
        clock = cyc / (rt * 10.0) if rt else 0.0                 # GHz: s_memrealtime ticks every 10 ns
        busy = nt * groups * 3 * 32.0 / cyc if cyc else 0.0
        ref = os.path.join('/tmp', 'wg_ref_%d_%d_%d.npy' % (HW, Ci, Co))
        d = dw.cpu().numpy()
        if tag == 'base': np.save(ref, d)
        err = ''
        if os.path.exists(ref) and (not tag.startswith('abl') or tag == 'abl_presplit'):
            r = np.load(ref)
            err = 'max|d-base|/max|base| %.1e' % (np.abs(d - r).max() / np.abs(r).max())
        tf = 2.0 * 9 * Ci * Co * N * HW * HW / us / 1e6
        out.append('%4d^2 %3d->%3d: %7.1f us %6.1f TF/s-alg | loop %8d cyc, clock %5.2f GHz, MFMA issue %4.1f %% of the loop | %s'
                   % (HW, Ci, Co, us, tf, cyc, clock, 100 * busy, err))
    print('%-22s' % tag + ('\n' + ' ' * 22).join(out), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        from scripts.build_variant import build_variant
        for tag, defs in VARIANTS:
            if len(sys.argv) > 2 and tag not in sys.argv[2:]: continue
            print(build_variant('wg_' + tag, ['wgrad_f16x3.hip'], ['-DDC_WG_CLOCK'] + defs), flush=True)
    elif sys.argv[1] == '--worker':
        worker(sys.argv[2])
    else:
        tags = sys.argv[2:] or [t for t, _ in VARIANTS]
        for tag in tags:
            subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', tag], check=False,
                           env=dict(os.environ, DC_LIB_PATH=lib_of(tag)))
