"""Are the level-0 (512^2) convolutions bound by their epilogue stores?  Times the training forward with BatchNorm partials of the
32 -> 32 layer (256-thread kernel) and of the 64 -> 32 layer (role-split <4,1,3>), and the 64-column layers for comparison, with the
shipped library and with an ablation build that issues no output stores (results garbage; the statistics keep the MFMAs alive).
    python scripts/level0_store_ablate.py build | run"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBS = [('shipped', None), ('no output stores', 'nostore'), ('shipped', None), ('no output stores', 'nostore')]
SHAPES = [(512, 32, 32), (512, 64, 32), (256, 64, 64), (128, 128, 128)]


def worker():
    import torch
    from deep_calcium_amd._lib import lib
    L = lib()
    N = 16
    out = []
    for HW, Ci, Co in SHAPES:
        x = torch.randn(N, HW, HW, Ci, device='cuda')
        K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
        wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
        L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
        z = torch.empty(N, HW, HW, Co, device='cuda')
        rows = L.dc_conv3x3_stats_rows(N, HW, HW, Ci, Co)
        stats = torch.zeros(rows * Co * 2, dtype=torch.float64, device='cuda')
        run = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr(), rows, None, None, 0,
                                             None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append('%d^2 %d->%d %.1f us' % (HW, Ci, Co, e0.elapsed_time(e1) * 50))
    print(' | '.join(out), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        from scripts.build_variant import build_variant
        print(build_variant('nostore', ['igemm_f16x3.hip', 'igemm_pp.hip'], ['-DDC_F_ABL=1', '-DDC_PP_ABL=64']))
    elif sys.argv[1] == '--worker':
        worker()
    else:
        for label, tag in LIBS:
            env = dict(os.environ)
            if tag:
                env['DC_LIB_PATH'] = os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_%s.so' % tag)
            print('%-18s' % label, end=' ', flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--worker'], env=env, stderr=subprocess.DEVNULL)
