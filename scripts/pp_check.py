"""Bit-for-bit A/B of the persistent role-split conv kernel (igemm_pp.hip) against the 256-thread kernel it replaces:
two processes (DC_IGEMM_PP=1 / 0) run the same launches and the outputs must be IDENTICAL; prints timings of both."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(2, 32, 32, 64, 64), (1, 64, 64, 128, 128), (2, 40, 40, 144, 96), (16, 256, 256, 64, 64), (16, 128, 128, 128, 128),
          (16, 64, 64, 256, 256), (16, 32, 32, 512, 512), (16, 256, 256, 128, 64), (3, 50, 70, 80, 72),
          (16, 512, 512, 32, 32), (16, 512, 512, 64, 32), (16, 256, 256, 32, 64), (2, 40, 40, 32, 32), (1, 64, 64, 48, 24),
          (2, 72, 40, 48, 96)]


def worker(out):
    import torch
    from deep_calcium_amd._lib import lib
    L = lib()
    res = {}
    for si, (N, H, W, Ci, Co) in enumerate(SHAPES):
        g = torch.Generator(device='cuda').manual_seed(si)
        x = torch.randn(N, H, W, Ci, device='cuda', generator=g)
        K = torch.randn(3, 3, Ci, Co, device='cuda', generator=g) * 0.05
        b = torch.randn(Co, device='cuda', generator=g)
        wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
        L.dc_pack_weights_f16x3(K.data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
        wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Co, Ci), device='cuda')
        L.dc_pack_weights_f16x3(K.data_ptr(), wpd.data_ptr(), 9, Co, Ci, Ci * Co, 1, Co, 1, None)
        tiles = L.dc_conv3x3_tiles(N, H, W, Co)
        stats = torch.zeros(tiles * Co * 2, dtype=torch.float64, device='cuda')
        z = torch.full((N, H, W, Co), float('nan'), device='cuda')
        sc = torch.rand(Ci, device='cuda', generator=g) + 0.5
        sh = torch.randn(Ci, device='cuda', generator=g) * 0.3
        ab = sc.abs() * 8 + sh.abs()
        flag = torch.zeros(4, device='cuda')
        dx = torch.full((N, H, W, Ci), float('nan'), device='cuda')
        dzs = torch.ones(1, device='cuda') * 4.0
        runs = {
            'fwd_stats': lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp.data_ptr(), b.data_ptr(), z.data_ptr(), Co, stats.data_ptr(), 0, None, None, 0, None, 0, None, 0, None, N, H, W, Ci, Co, None),
            'fwd_bnin': lambda: L.dc_conv3x3_fwd_bnin_f16x3(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), wp.data_ptr(), b.data_ptr(), z.data_ptr(), Co, stats.data_ptr(), 0, None, None, 0, None, N, H, W, Ci, Co, None),
            'infer': lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp.data_ptr(), None, z.data_ptr(), Co, None, 0, sc[:1].expand(Co).contiguous().data_ptr() if False else b.data_ptr(), b.data_ptr(), 1, None, 0, flag.data_ptr(), -1, None, N, H, W, Ci, Co, None),
            'dgrad': lambda: L.dc_conv3x3_dgrad_f16x3(z.data_ptr(), wpd.data_ptr(), dx.data_ptr(), dzs.data_ptr(), None, 0, None, N, H, W, Ci, Co, None),
        }
        for name, fn in runs.items():
            stats.zero_()
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res['%d_%s_t' % (si, name)] = np.float64(e0.elapsed_time(e1) / 3)
            res['%d_%s' % (si, name)] = (dx if name == 'dgrad' else z).cpu().numpy().copy()
            if 'stats' in name or 'bnin' in name:
                res['%d_%s_st' % (si, name)] = stats.cpu().numpy().copy()
        z.normal_(generator=g)          # dgrad input for the next comparison is deterministic anyway
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        worker(sys.argv[1])
        sys.exit(0)
    outs = []
    for pp in ('1', '0'):
        out = '/tmp/pp_check_%s.npz' % pp
        r = subprocess.run([sys.executable, os.path.abspath(__file__), out], env=dict(os.environ, DC_IGEMM_PP=pp), timeout=600)
        if r.returncode != 0:
            raise SystemExit('worker DC_IGEMM_PP=%s failed: %d' % (pp, r.returncode))
        outs.append(np.load(out))
    a, b = outs
    bad = 0
    for k in a.files:
        if k.endswith('_t'):
            continue
        same = np.array_equal(a[k], b[k], equal_nan=True)
        if not same:
            bad += 1
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print('MISMATCH', k, 'max abs diff', np.nanmax(d), 'nan pp/old', np.isnan(a[k]).sum(), np.isnan(b[k]).sum())
    for si, shp in enumerate(SHAPES):
        print(shp, '  '.join('%s %.0f/%.0f us' % (n, 1e3 * a['%d_%s_t' % (si, n)], 1e3 * b['%d_%s_t' % (si, n)])
                             for n in ('fwd_stats', 'fwd_bnin', 'infer', 'dgrad')), '(pp/old)')
    print('bitwise identical' if not bad else '%d arrays differ' % bad)
    sys.exit(1 if bad else 0)
