"""Model(device='cuda:1') while device 0 is current (needs 2 GPUs: skipped on the 1-GPU test box).  Every engine entry point
that launches -- forward_train, backward (its kernels, the per-device LDS-attribute / CU-count caches, the side-stream
events), adam_step -- and Model.train_on_device_batch's events / pinned copies must act on the ENGINE's device."""
import numpy as np
import pytest

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_train_step_on_non_current_device_equals_device_0():
    from deep_calcium_amd.model import Model, Adam
    N, H, W, nfb = 2, 64, 64, 8
    Wt = on.init_weights(nfb, seed=5, randomize_bn=True)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    outs = []
    torch.cuda.set_device(0)
    for dev in ('cuda:0', 'cuda:1'):
        m = Model((H, W), nfb, device=dev)
        m.compile(Adam(0.002), 'binary_crossentropy')
        m.set_weights(Wt)
        assert torch.cuda.current_device() == 0
        vals = [m.train_on_batch(x, y, drop_masks=masks) for _ in range(2)]
        torch.cuda.synchronize(dev)
        assert torch.cuda.current_device() == 0
        outs.append((vals, m.engine.pflat.cpu().numpy().copy(), m.engine.gflat.cpu().numpy().copy(), m.predict(x)))
    (v0, p0, g0, q0), (v1, p1, g1, q1) = outs
    assert v0 == v1 and np.array_equal(p0, p1) and np.array_equal(g0, g1) and np.array_equal(q0, q1)
