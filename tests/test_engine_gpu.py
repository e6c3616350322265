"""Whole-network GPU parity: UNetEngine (HIP kernels through the C ABI) vs the float64 oracle.

Targets from BASELINE.md section 4: probabilities max-abs <= 1e-4, BCE loss <= 1e-4, pool argmax bit-exact,
thresholded masks identical away from |p - 0.5| < 1e-4.
"""
import numpy as np
import pytest

from oracle import unet_numpy as on
from _forced import assert_forcing_is_benign, device_decisions, grad_report, count_flips

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

# Gradient bound with the oracle routed through the device's own ReLU gates and pool indices (tests/_forced.py): what is
# left is fp32-vs-float64 rounding.  Per tensor: max|g - r| <= GRAD_TOL * max|r|; whole gradient: rel-L2 <= GRAD_TOL.
GRAD_TOL = 1e-4


MODES = ['f16x3', 'f32']


def make_engine(H, W, nfb, drp=0.25, randomize_bn=True, seed=7535, mfma='f16x3'):
    from deep_calcium_amd.net import UNetEngine
    eng = UNetEngine((H, W), nb_filters_base=nfb, prop_dropout_base=drp, mfma=mfma)
    Wt = on.init_weights(nfb, seed=seed, randomize_bn=randomize_bn)
    eng.set_weights(Wt)
    return eng, Wt


def dev_masks(masks):
    return {k: torch.from_numpy(v).cuda() for k, v in masks.items()}


@pytest.mark.parametrize('mfma', MODES)
@pytest.mark.parametrize('N,H,W,nfb', [(2, 32, 32, 32), (1, 64, 64, 8), (3, 96, 96, 4), (1, 48, 80, 16)])
def test_forward_inference_matches_oracle(N, H, W, nfb, mfma):
    eng, Wt = make_engine(H, W, nfb, mfma=mfma)
    x, _ = on.synthetic_batch(N, H, W)
    p_ref = on.UNetOracle(Wt, nfb).forward(x, training=False)
    p = eng.forward_infer(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(p - p_ref).max() < 1e-4
    away = np.abs(p_ref - 0.5) >= 1e-4
    assert np.array_equal((p > 0.5)[away], (p_ref > 0.5)[away])
    # get_weights round trip (Keras order, 134 arrays at any nfb)
    got = eng.get_weights()
    assert len(got) == 134 and all(np.array_equal(a, np.asarray(b, np.float32)) for a, b in zip(got, Wt))


@pytest.mark.parametrize('mfma', MODES)
# (2, 96, 96, 32) / (2, 128, 128, 32): the reference's own training windows (fit() default 96 x 96; examples/neurons/unet2ds_nf.py 128 x 128)
@pytest.mark.parametrize('N,H,W,nfb', [(1, 16, 16, 4), (2, 32, 32, 32), (2, 64, 64, 8), (2, 64, 64, 32), (2, 96, 96, 32), (2, 128, 128, 32)])
def test_train_forward_backward_matches_oracle(N, H, W, nfb, mfma):
    eng, Wt = make_engine(H, W, nfb, mfma=mfma)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    orc = on.UNetOracle(Wt, nfb)
    taps, cache = {}, {}
    orc.forward(x, True, masks, cache=cache, taps=taps)
    loss_ref, p_ref, G_ref, stats_ref = orc.loss_and_grads(x, y, masks)

    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p = eng.forward_train(xd, yd, dev_masks(masks), update_moving=False).cpu().numpy()
    sums = eng.read_sums()
    assert np.abs(p - p_ref).max() < 1e-4
    assert abs(sums[0] / p.size - loss_ref) < 1e-4
    # pool argmax indices: bit-exact wherever the window has a clear winner (or an exact tie) in the oracle
    A = eng._acts(N)
    for lvl in range(4):
        idx = A['idx%d' % lvl].cpu().numpy()
        ref = taps['p%d_idx' % lvl]
        src = taps[('e%db' % lvl)]
        win = np.stack([src[:, 0::2, 0::2], src[:, 0::2, 1::2], src[:, 1::2, 0::2], src[:, 1::2, 1::2]], -1)
        srt = np.sort(win, -1)
        clear = (srt[..., -1] - srt[..., -2] > 1e-4) | (srt[..., -1] == srt[..., -2])
        assert np.array_equal(idx[clear], ref[clear])
        assert (idx != ref).mean() < 1e-3
    bs = eng.batch_stats()
    for name, (mu, var) in stats_ref.items():
        assert np.abs(bs[name][0] - mu).max() < 1e-4 * max(1.0, np.abs(mu).max()), name
        assert np.abs(bs[name][1] - 1 / np.sqrt(var + 1e-3)).max() < 1e-4 * (1 / np.sqrt(var + 1e-3)).max(), name

    eng.backward()
    G = eng.grads()
    # ---- the tight comparison: same gates, same pool routes -> pure rounding ------------------------------------------
    dec = device_decisions(eng, N)
    # ReLU gates that differ from the un-forced float64 run (pre-activation within fp32 rounding of 0): each moves one dz
    # element by O(|da|) -- a legitimate discontinuity, which is why the un-forced comparison below stays in loose norms
    flips = count_flips(dec, cache, masks)
    loss_f, p_f, G_f, _ = on.UNetOracle(Wt, nfb, force=dec).loss_and_grads(x, y, masks)
    assert np.abs(p_f - p_ref).max() < 1e-6 and abs(loss_f - loss_ref) < 1e-6      # forcing moves the function by rounding only
    worst, rel, _ = grad_report(G, G_f, '%s N%d %dx%d nfb%d forced gates: ' % (mfma, N, H, W, nfb))
    assert worst < GRAD_TOL and rel < GRAD_TOL, (worst, rel)
    grad_report(G, G_ref, '%s N%d %dx%d nfb%d un-forced (%d gate flips): ' % (mfma, N, H, W, nfb, flips))
    flat_g, flat_r = [], []
    for name, ref in G_ref.items():
        for j, (g, r) in enumerate(zip(G[name], ref)):
            r = r.reshape(g.shape)
            if j == 1 and name != 'out':
                # conv bias in front of BatchNorm: analytically zero gradient, pure rounding noise on both sides
                assert np.abs(g).max() < 1e-5, (name, np.abs(g).max())
                continue
            flat_g.append(g.ravel().astype(np.float64))
            flat_r.append(r.ravel())
            scale = max(np.abs(r).max(), 1e-6)
            if flips == 0:
                assert np.abs(g - r).max() < 2e-3 * scale, (name, j, np.abs(g - r).max(), scale)
            else:
                # a flipped ReLU gate perturbs a few hundred entries of one tensor by O(1e-2) of its scale
                assert np.linalg.norm(g - r) < 0.1 * np.linalg.norm(r) + 1e-7, (name, j, flips)
    fg, fr = np.concatenate(flat_g), np.concatenate(flat_r)
    cos = fg.dot(fr) / (np.linalg.norm(fg) * np.linalg.norm(fr))
    assert cos > 0.9995, (cos, flips)
    print('relu gate flips: %d, gradient cosine %.7f' % (flips, cos))


def test_upsampling_branch_matches_oracle():
    """unet(upsampling_or_transpose='upsampling') (unet_2d_summary.py:160-161): UpSampling2D + dropout instead of the
    transposed conv; 110 weight arrays, 3c-channel concat."""
    from deep_calcium_amd.net import UNetEngine
    N, H, W, nfb = 2, 32, 32, 8
    eng = UNetEngine((H, W), nfb, upsampling=True)
    Wt = on.init_weights(nfb, randomize_bn=True, upsampling=True)
    assert len(Wt) == 110 and len(eng.weight_shapes()) == 110
    eng.set_weights(Wt)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W, upsampling=True)
    orc = on.UNetOracle(Wt, nfb, upsampling=True)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    assert np.abs(eng.forward_infer(xd).cpu().numpy() - orc.forward(x)).max() < 1e-4
    p = eng.forward_train(xd, yd, dev_masks(masks), update_moving=False).cpu().numpy()
    eng.backward()
    G = eng.grads()
    loss_ref, p_ref, G_ref, _ = on.UNetOracle(Wt, nfb, upsampling=True, force=device_decisions(eng, N)).loss_and_grads(x, y, masks)
    assert np.abs(p - p_ref).max() < 1e-4 and abs(eng.read_sums()[0] / p.size - loss_ref) < 1e-4
    assert_forcing_is_benign(Wt, nfb, x, masks, p_ref, p_dev=p, upsampling=True)
    worst, rel, _ = grad_report(G, G_ref, 'upsampling branch, forced gates: ')
    assert worst < GRAD_TOL and rel < GRAD_TOL, (worst, rel)
    # random-RNG dropout path runs and is reproducible in backward (same seed regenerates the masks)
    eng.forward_train(xd, yd, None, update_moving=False)
    eng.backward()
    assert np.isfinite(eng.gflat.cpu().numpy()).all()


@pytest.mark.parametrize('loss', ['weighted_binary_crossentropy', 'dice_loss', 'dicesq_loss'])
def test_alternate_losses_match_oracle(loss):
    from deep_calcium_amd.model import LOSS_KINDS, metrics_from_sums
    N, H, W, nfb = 2, 32, 32, 8
    eng, Wt = make_engine(H, W, nfb)
    eng.loss_kind = LOSS_KINDS[loss]
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    eng.forward_train(xd, yd, dev_masks(masks), update_moving=False)
    eng.backward()
    G = eng.grads()
    loss_ref, p_ref, G_ref, _ = on.UNetOracle(Wt, nfb, force=device_decisions(eng, N)).loss_and_grads(x, y, masks, loss=loss)
    assert abs(metrics_from_sums(eng.read_sums(), N * H * W, loss)['loss'] - loss_ref) < 1e-4
    assert_forcing_is_benign(Wt, nfb, x, masks, p_ref)
    worst, rel, _ = grad_report(G, G_ref, '%s, forced gates: ' % loss)
    assert worst < GRAD_TOL and rel < GRAD_TOL, (worst, rel)


@pytest.mark.parametrize('mfma', MODES)
def test_train_steps_loss_matches_oracle(mfma):
    """Two full train steps (fwd + bwd + Keras Adam + BN moving stats), explicit dropout masks: BCE loss and
    the next-step probabilities stay within 1e-4 of the float64 oracle."""
    N, H, W, nfb = 2, 32, 32, 16
    eng, Wt = make_engine(H, W, nfb, randomize_bn=False, mfma=mfma)
    orc = on.UNetOracle(Wt, nfb)
    state = dict(it=0, m={}, v={})
    x, y = on.synthetic_batch(N, H, W)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    for step in range(2):
        masks = on.make_drop_masks(nfb, N, H, W, seed=7 + step)
        loss_ref, _ = orc.train_step(x, y, state, masks, lr=0.002)
        eng.forward_train(xd, yd, dev_masks(masks))
        eng.backward()
        eng.adam_step(0.002)
        loss = eng.read_sums()[0] / (N * H * W)
        assert abs(loss - loss_ref) < 1e-4, (step, loss, loss_ref)
    # moving statistics followed the Keras rule (biased variance, momentum 0.99 / 0.5)
    W_hip, W_ref = eng.get_weights(), orc.weights()
    shapes = on.weight_shapes(nfb)
    k = 0
    for name, kind, cin, cout, mom in on.layer_table(nfb):
        n = 2 if kind == 'head' else 6
        if kind != 'head':
            assert np.abs(W_hip[k + 4] - W_ref[k + 4]).max() < 1e-4, name
            assert np.abs(W_hip[k + 5] - W_ref[k + 5]).max() < 1e-4 * max(1, np.abs(W_ref[k + 5]).max()), name
            # Adam's first steps are sign-like (|delta| ~ lr whatever |g|): entries whose gradient is rounding
            # noise may move the other way, everything else must move together
            assert np.mean(np.abs(W_hip[k] - W_ref[k]) > 5e-4) < 0.02, name
        k += n
    p_ref = orc.forward(x, training=False)
    p = eng.forward_infer(xd).cpu().numpy()
    assert np.abs(p - p_ref).max() < 5e-3      # after 2 Adam steps of size lr the sign-like update amplifies 1e-7 noise


@pytest.mark.parametrize('mfma', MODES)
def test_determinism_same_input_twice(mfma):
    """Bit-identical gradients across two runs: no atomics anywhere (wgrad slabs / BN partials are fixed-order)."""
    N, H, W, nfb = 2, 32, 32, 32
    eng, _ = make_engine(H, W, nfb, mfma=mfma)
    x, y = on.synthetic_batch(N, H, W)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    outs = []
    for _ in range(2):
        eng.forward_train(xd, yd, None, update_moving=False)
        eng.backward()
        torch.cuda.synchronize()
        outs.append(eng.gflat.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])
