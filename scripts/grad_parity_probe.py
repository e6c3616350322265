"""Gradient parity with the oracle run through the device's own ReLU gates / pool indices (tests/_forced.py): prints the
per-tensor and whole-gradient errors, forced and un-forced, for a list of shapes.  python scripts/grad_parity_probe.py [full]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from _forced import device_decisions, grad_report, count_flips     # noqa: E402
from deep_calcium_amd.net import UNetEngine                        # noqa: E402
from oracle import unet_numpy as on                                # noqa: E402
from oracle.unet_torch import UNetTorch                            # noqa: E402

shapes = [(1, 16, 16, 4), (2, 32, 32, 32), (2, 64, 64, 8), (2, 128, 128, 32), (4, 96, 96, 16)]
if 'full' in sys.argv:
    shapes = [(2, 512, 512, 32)]
for (N, H, W, nfb) in shapes:
    for mode in ('f16x3', 'f32'):
        Wt = on.init_weights(nfb, seed=77, randomize_bn=True)
        x, y = on.synthetic_batch(N, H, W)
        masks = on.make_drop_masks(nfb, N, H, W)
        eng = UNetEngine((H, W), nfb, mfma=mode)
        eng.set_weights(Wt)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        md = {k: torch.from_numpy(v).cuda() for k, v in masks.items()}
        p = eng.forward_train(xd, yd, md, update_moving=False).cpu().numpy()
        eng.backward()
        G = eng.grads()
        dec = device_decisions(eng, N)
        t0 = time.time()
        big = H >= 256
        if big:
            lf, pf, Gf, _ = UNetTorch(Wt, nfb, force=dec).loss_and_grads(x, y, masks)
            l0, p0, G0, _ = UNetTorch(Wt, nfb).loss_and_grads(x, y, masks)
            flips = -1
        else:
            cache = {}
            on.UNetOracle(Wt, nfb).forward(x, True, masks, cache=cache)
            flips = count_flips(dec, cache, masks)
            lf, pf, Gf, _ = on.UNetOracle(Wt, nfb, force=dec).loss_and_grads(x, y, masks)
            l0, p0, G0, _ = on.UNetOracle(Wt, nfb).loss_and_grads(x, y, masks)
        print('== N%d %dx%d nfb%d %s: flips %d, |p-pf| %.2e, |pf-p0| %.2e  (oracle %.1fs)' %
              (N, H, W, nfb, mode, flips, np.abs(p - pf).max(), np.abs(pf - p0).max(), time.time() - t0))
        grad_report(G, Gf, '  forced:   ')
        grad_report(G, G0, '  unforced: ')
        del eng
        torch.cuda.empty_cache()
