"""End-to-end fit() speed on synthetic Neurofinder-like datasets (the reference's default shapes: 128x128 training
windows, batch 20, 512x512 validation): steps/s of the whole Python path (RNG-exact batch generator thread + HIP step)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import UNet2DSummary

tmp = tempfile.mkdtemp()
rs = np.random.RandomState(0)
paths = []
for k in range(4):
    hw = (512, 512)
    mean = (rs.random_sample(hw) * 500 + 50).astype(np.float16)
    masks = np.zeros((30,) + hw, np.int8)
    for z in range(30):
        cy, cx = rs.randint(8, hw[0] - 8), rs.randint(8, hw[1] - 8)
        masks[z, cy - 4:cy + 5, cx - 4:cx + 5] = 1
    p = os.path.join(tmp, 'ds%d.npz' % k)
    np.savez(p, series_mean=mean, masks_raw=masks, name=np.array('neurofinder.0%d.00' % k))
    paths.append(p)
np.random.seed(1)
model = UNet2DSummary(cpdir=os.path.join(tmp, 'cp'))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
hist, _ = model.fit(paths, shape_trn=(128, 128), shape_val=(512, 512), batch_size_trn=20, nb_steps_trn=steps, nb_epochs=2,
                    keras_callback_verbosity=0) if 'keras_callback_verbosity' in UNet2DSummary.fit.__code__.co_varnames else \
    model.fit(paths, shape_trn=(128, 128), shape_val=(512, 512), batch_size_trn=20, nb_steps_trn=steps, nb_epochs=2)
dt = time.time() - t0
print('fit: 2 epochs x %d steps of batch 20 @128x128 (+ validation) in %.2f s -> %.1f steps/s, %.0f windows/s' % (steps, dt, 2 * steps / dt, 40 * steps / dt))
