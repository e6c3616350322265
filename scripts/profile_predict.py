"""Host-side profile of UNet2DSummary.predict (BASELINE configs[4]): where the time outside the kernels goes.
    python scripts/profile_predict.py [--tta]"""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from deep_calcium_amd import UNet2DSummary, unet_hip      # noqa: E402

tmp = tempfile.mkdtemp()
rs = np.random.RandomState(1)
paths = []
for k in range(19):
    p = os.path.join(tmp, 'ds%02d.npz' % k)
    np.savez(p, series_mean=(rs.random_sample((512, 512)) * 900 + 100).astype(np.float16), name=np.array('nf.%02d' % k))
    paths.append(p)
mpath = os.path.join(tmp, 'model.hdf5')
unet_hip((512, 512)).save(mpath)
api = UNet2DSummary(cpdir=tmp)
aug = '--tta' in sys.argv
api.predict(paths, mpath, augmentation=aug)
pr = cProfile.Profile()
pr.enable()
api.predict(paths, mpath, augmentation=aug)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
