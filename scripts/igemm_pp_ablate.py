"""What stretches a step of the role-split conv kernel beyond its 108 x 32 = 3 456 cycles of MFMA issue?  Builds
igemm_pp.hip with -DDC_IGEMM_TRACE -DDC_PP_ABL=<mask> (bit 0 no epilogue, 1 no split / LDS writes, 2 fragments read once
per step, 3 no global loads; results are garbage) and reports the kernel time and the consumers' MFMA-step cycles.
    python scripts/igemm_pp_ablate.py build [masks...]                 # HERE: the .so files travel to the GPU box
    python scripts/igemm_pp_ablate.py HW Cin Cout variant [masks...]   # variant 0: forward, no BatchNorm partials (<2,2,0>);
                                                                       # 3: forward + per-workgroup partials (<2,2,3>);
                                                                       # 1: data gradient + BatchNorm-backward sums (<2,2,1>)"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'deep_calcium_amd', 'csrc')
from deep_calcium_amd._build import SOURCES            # noqa: E402


def build(mask):
    lib = os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_ppabl%d.so' % mask)
    if not os.path.exists(lib) or any(os.path.getmtime(os.path.join(CSRC, f)) > os.path.getmtime(lib) for f in os.listdir(CSRC)):
        from scripts.build_variant import build_variant
        build_variant('ppabl%d' % mask, ['igemm_pp.hip'], ['-DDC_IGEMM_TRACE', '-DDC_PP_ABL=%d' % mask])
    return lib


def make_run(L, variant, HW, Ci, Co, N=16, presplit=False):
    """One conv3x3 launch of the role-split kernel as a closure: variant 0 forward without BatchNorm partials (<*,*,0>), 3 forward
    with per-workgroup partial rows (<*,*,3>), 1 data gradient that also emits BatchNorm-backward sums (<*,*,1>).  Returns
    (run, keep-alive tensors)."""
    import torch
    x = torch.randn(N, HW, HW, Ci, device='cuda')
    if presplit:                       # -DDC_PP_ABL=32: the same values stored as fp16 (hi, lo) pairs
        from scripts._presplit import presplit_pack
        x = presplit_pack(x)
    K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
    wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    z = torch.empty(N, HW, HW, Co, device='cuda')
    keep = [x, K, wp16, z]
    if variant == 3:
        rows = L.dc_conv3x3_stats_rows(N, HW, HW, Ci, Co)
        stats = torch.zeros(rows * Co * 2, dtype=torch.float64, device='cuda')
        keep.append(stats)
        run = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr(), rows, None, None, 0,
                                             None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
    elif variant == 0:
        run = lambda: L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, None, 0, None, None, 0,
                                             None, 0, None, 0, None, N, HW, HW, Ci, Co, None)
    else:           # the data gradient of a Cout -> Cin layer that also emits the sums of the BatchNorm layer it writes `da` of
        assert Ci == Co, 'variant 1 is timed on square layers (the forward packing stands in for the flipped one)'
        rows = L.dc_conv3x3_dgrad_bnred_blocks(N, HW, HW, Ci, Co)
        assert rows > 0
        zin = torch.randn(N, HW, HW, Ci, device='cuda')
        mean, invstd, gamma, beta = (torch.rand(Ci, device='cuda') + 0.5 for _ in range(4))
        part = torch.zeros(rows * Ci * 2, device='cuda'); amax = torch.zeros(rows * Ci, device='cuda')
        keep += [zin, mean, invstd, gamma, beta, part, amax]
        run = lambda: L.dc_conv3x3_dgrad_bnred_f16x3(x.data_ptr(), wp16.data_ptr(), z.data_ptr(), None, None, 0, zin.data_ptr(),
                                                     mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     part.data_ptr(), amax.data_ptr(), N, HW, HW, Ci, Co, None)
    return run, keep


def worker(lib_path, HW, Ci, Co, mask=0, variant=3):
    os.environ['DC_LIB_PATH'] = lib_path
    import torch
    from deep_calcium_amd._lib import lib
    L = lib()
    N = 16
    run, keep = make_run(L, variant, HW, Ci, Co, N, presplit=bool(mask & 32))
    trace = torch.zeros(8 * 3 * 512, dtype=torch.int64, device='cuda')
    fn = L.cdll.dc_debug_set_pp_trace
    fn.argtypes = [ctypes.c_void_p]
    fn(None)
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fn(trace.data_ptr()); run(); torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 3, 512)
    nch = Ci // 16
    if mask & 16:            # detail stamps: consumers emit [tile setup done, first fragments landed, done, released] per step
        seg = {'setup (decode, zero acc)': [], 'first fragments (LDS latency)': [], 'MFMA block': [], 'barrier wait': []}
        for cta in range(t.shape[0]):
            for r in (0, 1):
                n = int(t[cta, r, 0])
                if n < 6: continue
                P = t[cta, r, 1:n + 1].astype(np.float64)
                rel0, body = P[1], P[2:]
                steps = len(body) // 4
                body = body[:steps * 4].reshape(steps, 4)
                prev = np.concatenate([[rel0], body[:-1, 3]])
                k = np.arange(steps)
                mine = ((k // nch) % 2) == r
                for st in np.where(mine)[0]:
                    s1, s2, d, rl = body[st]
                    seg['setup (decode, zero acc)'].append(s1 - prev[st]); seg['first fragments (LDS latency)'].append(s2 - s1)
                    seg['MFMA block'].append(d - s2); seg['barrier wait'].append(rl - d)
        print('%7.1f us | ' % us + ' | '.join('%s %5.0f' % (k_, np.median(v)) for k_, v in seg.items()) + '   (cycles, medians over MFMA steps)')
        return
    mf, ep, pr, per = [], [], [], []
    for cta in range(t.shape[0]):
        n = int(t[cta, 2, 0])
        if n == 0: continue
        P = t[cta, :, 1:n + 1].astype(np.float64)
        done, rel = P[:, 0::2], P[:, 1::2]
        steps = min(done.shape[1], rel.shape[1])
        work = done[:, 1:steps] - rel[:, 0:steps - 1]
        k = np.arange(steps - 1) + 1
        for r in (0, 1):
            mine = ((k // nch) % 2) == r
            mf += list(work[r][mine]); ep += list(work[r][~mine])
        pr += list(work[2]); per += list(rel[0, 1:steps] - rel[0, 0:steps - 1])
    print('%7.1f us | step period %5.0f | consumer MFMA step %5.0f | epilogue step %5.0f | producer step %5.0f   (cycles, medians)' % (
        us, np.median(per), np.median(mf), np.median(ep), np.median(pr)))


if __name__ == '__main__':
    if sys.argv[1] == '--worker':
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
        sys.exit(0)
    if sys.argv[1] == 'build':
        for m in [int(v) for v in sys.argv[2:]] or [0, 2, 4, 8, 10, 14, 16]:
            print(build(m), flush=True)
        sys.exit(0)
    HW, Ci, Co, variant = sys.argv[1:5]
    masks = [int(m) for m in sys.argv[5:]] or [0, 2, 4, 8, 10, 14, 16]
    names = {0: 'full kernel', 1: 'no epilogue', 2: 'no split / LDS writes', 4: 'fragments read once per step', 8: 'no global loads',
             16: 'detail stamps', 32: 'raw bits instead of the fp16 split (a pre-split operand)'}
    for m in masks:
        lib = build(m)
        label = ' + '.join(names[b] for b in (1, 2, 4, 8, 16, 32) if m & b) or names[0]
        print('%-60s' % label, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', lib, HW, Ci, Co, str(m), variant], check=True,
                       env=dict(os.environ, DC_LIB_PATH=lib))      # read when the package is imported
