"""BASELINE.json configs[1] and configs[4] at their REAL size (nb_filters_base 32, 512 x 512) under `-m gpu`:

  configs[1]  forward only, batch 8: every image of the batch-8 forward equals its batch-1 forward bit for bit, one of them
              is held against the float64 torch oracle (1e-4);
  configs[4]  UNet2DSummary.predict(augmentation=True) (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:585-595
              with the table of utils/neurons.py:112-137): the device's batched 8x test-time augmentation -- gather the 8
              copies, ONE batch-8 forward, inverse-map, mean -- against 8 float64 oracle forwards mapped back with the
              reference table on the host: mean probability within 1e-4, masks equal away from the threshold; and the
              19-dataset pipelined path equals the one-by-one path at full size.
Synthetic images (the Neurofinder data and the released weights are not available offline)."""
import numpy as np
import pytest

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

NFB, H, W = 32, 512, 512


@pytest.fixture(scope='module')
def engine():
    from deep_calcium_amd.net import UNetEngine
    Wt = on.init_weights(NFB, seed=77, randomize_bn=True)
    eng = UNetEngine((H, W), nb_filters_base=NFB)
    eng.set_weights(Wt)
    return eng, Wt


def test_configs1_forward_batch8_is_batch_independent_and_matches_oracle(engine):
    from oracle.unet_torch import UNetTorch
    eng, Wt = engine
    x, _ = on.synthetic_batch(8, H, W, seed_x=31)
    xd = torch.from_numpy(x).cuda()
    p8 = eng.forward_infer(xd).clone()
    for i in range(8):
        p1 = eng.forward_infer(xd[i:i + 1].contiguous())
        assert torch.equal(p1[0], p8[i]), i
    ref = UNetTorch(Wt, NFB, dtype=torch.float64, requires_grad=False)
    with torch.no_grad():
        p_ref = ref.forward(x[5:6], training=False).detach().numpy()[0]
    got = p8[5].cpu().numpy()
    assert np.abs(got - p_ref).max() < 1e-4
    away = np.abs(p_ref - 0.5) >= 1e-4
    assert np.array_equal((got > 0.5)[away], (p_ref > 0.5)[away])


def test_configs4_tta_mean_matches_eight_float64_oracle_forwards(engine):
    from deep_calcium_amd.unet2ds import INVERTIBLE_2D_AUGMENTATIONS
    from oracle.unet_torch import UNetTorch
    eng, Wt = engine
    rs = np.random.RandomState(5)
    hs, ws = 498, 507                                        # a dataset smaller than the window: reflect-padded, cropped back
    s = rs.standard_normal((hs, ws)).astype(np.float32)
    img = np.pad(s, ((0, H - hs), (0, W - ws)), mode='reflect')
    thr = 0.5
    mask, mean = eng.predict_tta(img, INVERTIBLE_2D_AUGMENTATIONS, hs, ws, thr, return_mean=True)
    assert not eng.infer_measured                            # normalised input: the optimistic range mode held
    ref = UNetTorch(Wt, NFB, dtype=torch.float64, requires_grad=False)
    mp = np.zeros((hs, ws))
    with torch.no_grad():
        for _, aug, inv in INVERTIBLE_2D_AUGMENTATIONS:      # unet_2d_summary.py:588-592, one forward per variant
            pr = ref.forward(np.ascontiguousarray(aug(img[np.newaxis])), training=False).detach().numpy()
            mp += inv(pr)[0, :hs, :ws] / len(INVERTIBLE_2D_AUGMENTATIONS)
    assert mean.shape == (hs, ws) and np.abs(mean - mp).max() < 1e-4, np.abs(mean - mp).max()
    away = np.abs(mp - thr) >= 1e-4
    assert away.mean() > 0.99 and np.array_equal(mask[away], (mp > thr).astype(np.uint8)[away])


def test_configs4_nineteen_dataset_pipeline_equals_one_by_one(engine):
    from deep_calcium_amd.unet2ds import INVERTIBLE_2D_AUGMENTATIONS
    eng, Wt = engine
    rs = np.random.RandomState(19)
    sizes = [(512, 512)] * 10 + [(int(rs.randint(400, 513)), int(rs.randint(400, 513))) for _ in range(9)]
    imgs = []
    for hs, ws in sizes:
        s = rs.standard_normal((hs, ws)).astype(np.float32)
        imgs.append(np.pad(s, ((0, H - hs), (0, W - ws)), mode='reflect'))
    job = eng.tta_begin(len(sizes), INVERTIBLE_2D_AUGMENTATIONS)
    for i, ((hs, ws), im) in enumerate(zip(sizes, imgs)):
        job.enqueue(i, im, hs, ws, 0.5)
    got = job.finish()
    assert len(got) == 19
    for i, ((hs, ws), im) in enumerate(zip(sizes, imgs)):
        want = eng.predict_tta(im, INVERTIBLE_2D_AUGMENTATIONS, hs, ws, 0.5)
        assert got[i].shape == (hs, ws) and np.array_equal(got[i], want), i
    assert 0.0 < np.mean([g.mean() for g in got]) < 1.0
