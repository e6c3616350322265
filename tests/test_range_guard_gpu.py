"""fp16 range guard of the split-fp16 (f16x3) contractions, and BatchNorm statistics at |mean| >> sigma.

The fp32 reference (Keras/TF conv + BatchNormalization, unet_2d_summary.py:163-167) has no exponent-range cliff:
whatever kernels `set_weights()` installs and whatever magnitude the activations reach, the result is an fp32 number.
The f16x3 path must match the float64 oracle at the usual 2e-5 for kernels scaled by 2^-12 / 2^8, for activations
beyond fp16's 65504, and for un-normalised inference inputs -- never return inf.
"""
import numpy as np
import pytest

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

_KEEP = []


@pytest.fixture(autouse=True)
def _keep_alive():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    return t


def rel_err(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / (np.abs(ref).max() + 1e-300))


def pack16(L, K, taps, Kdim, Ncols, s_tap, s_k, s_n, flip):
    dst = torch.empty(L.dc_pack_weights_f16x3_floats(taps, Kdim, Ncols), device='cuda')
    L.dc_pack_weights_f16x3(dev(K).data_ptr(), dst.data_ptr(), taps, Kdim, Ncols, s_tap, s_k, s_n, flip, None)
    return dst


@pytest.mark.parametrize('wscale', [2.0 ** -12, 1.0, 2.0 ** 8, 1e-6, 3e4])
def test_kernel_magnitude_does_not_matter(dclib, wscale):
    """conv3x3 forward / dgrad and conv-transpose forward with kernels of magnitude ~0.03 x wscale: the pack-time
    power-of-two scale keeps the hi/lo split inside fp16's normal range (unscaled, 2^-12 kernels would keep ~11 bits)."""
    L = dclib
    N, H, W, Ci, Co = 2, 24, 40, 64, 96
    rs = np.random.RandomState(3)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci)) * wscale).astype(np.float32)
    dz = rs.standard_normal((N, H, W, Co)).astype(np.float32)
    z_ref = on.conv3x3_fwd(x.astype(np.float64), K.astype(np.float64), np.zeros(Co))
    dx_ref, _, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    wp = pack16(L, K, 9, Ci, Co, Ci * Co, Co, 1, 0)
    wpd = pack16(L, K, 9, Co, Ci, Ci * Co, 1, Co, 1)
    torch.cuda.synchronize()
    ws = float(wp[-4].item())
    assert ws == 2.0 ** (10 - np.floor(np.log2(np.abs(K).max())))          # trailer = the exact power-of-two scale
    z = torch.full((N, H, W, Co), float('nan'), device='cuda')
    dx = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_conv3x3_fwd_f16x3(dev(x).data_ptr(), wp.data_ptr(), None, z.data_ptr(), Co, None, 0, None, None, 0, None, 0, None, 0, None,
                           N, H, W, Ci, Co, None)
    L.dc_conv3x3_dgrad_f16x3(dev(dz).data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None, 0, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z.cpu().numpy(), z_ref) < 2e-5
    assert rel_err(dx.cpu().numpy(), dx_ref) < 2e-5
    KT = (rs.standard_normal((2, 2, Co, Ci)) * np.sqrt(2.0 / (4 * Co)) * wscale).astype(np.float32)
    wpt = pack16(L, KT, 1, Ci, 4 * Co, 0, 1, Ci, 0)
    t = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    L.dc_convT2x2_fwd_f16x3(dev(x).data_ptr(), wpt.data_ptr(), None, t.data_ptr(), Co, None, None, None, 0, None, 0, None, 0,
                            N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(t.cpu().numpy(), on.convT2x2_fwd(x.astype(np.float64), KT.astype(np.float64), np.zeros(Co))) < 2e-5


def test_all_zero_kernel_packs_and_runs(dclib):
    L = dclib
    N, H, W, Ci, Co = 1, 16, 16, 32, 32
    wp = pack16(L, np.zeros((3, 3, Ci, Co), np.float32), 9, Ci, Co, Ci * Co, Co, 1, 0)
    z = torch.full((N, H, W, Co), float('nan'), device='cuda')
    L.dc_conv3x3_fwd_f16x3(dev(np.ones((N, H, W, Ci), np.float32)).data_ptr(), wp.data_ptr(), None, z.data_ptr(), Co, None, 0,
                           None, None, 0, None, 0, None, 0, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert float(wp[-4].item()) == 1.0 and (z == 0).all()


@pytest.mark.parametrize('amag', [3e5, 1e-6, 1.0])
def test_activation_magnitude_guard(dclib, amag):
    """Activations up to amag (beyond fp16's 65504 for 3e5; deep in the subnormals for 1e-6): with the producer's
    per-channel bound the forward conv, the conv-transpose and both weight gradients match the oracle; WITHOUT it the
    large case overflows -- the guard is load-bearing."""
    L = dclib
    N, H, W, Ci, Co = 2, 16, 24, 64, 64
    rs = np.random.RandomState(8)
    a = (np.maximum(rs.standard_normal((N, H, W, Ci)), 0) * amag / 4).astype(np.float32)
    a[0, 3, 5, 7] = amag
    bound = np.full(Ci, np.abs(a).max() * 1.5, np.float32)                   # any valid upper bound will do
    bound[1] = np.abs(a).max() * 40
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    dz = (rs.standard_normal((N, H, W, Co)) * 1e-6).astype(np.float32)
    z_ref = on.conv3x3_fwd(a.astype(np.float64), K.astype(np.float64), np.zeros(Co))
    _, dK_ref, _ = on.conv3x3_bwd(a.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    wp = pack16(L, K, 9, Ci, Co, Ci * Co, Co, 1, 0)
    ad, bd = dev(a), dev(bound)
    z = torch.full((N, H, W, Co), float('nan'), device='cuda')
    L.dc_conv3x3_fwd_f16x3(ad.data_ptr(), wp.data_ptr(), None, z.data_ptr(), Co, None, 0, None, None, 0, bd.data_ptr(), 0, None, 0, None,
                           N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.isfinite(z).all() and rel_err(z.cpu().numpy(), z_ref) < 2e-5
    scl = torch.empty(1, device='cuda')
    L.dc_pow2_scale_from_absmax(dev(np.array([np.abs(dz).max()], np.float32)).data_ptr(), 1, 1024.0, scl.data_ptr(), None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    dw = torch.full((3, 3, Ci, Co), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_f16x3(ad.data_ptr(), dev(dz).data_ptr(), dw.data_ptr(), ws.data_ptr(), scl.data_ptr(), bd.data_ptr(),
                             N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all() and rel_err(dw.cpu().numpy(), dK_ref) < 2e-5
    # conv-transpose: forward and weight gradient (the activation is the B operand there)
    KT = (rs.standard_normal((2, 2, Co, Ci)) * np.sqrt(2.0 / (4 * Co))).astype(np.float32)
    wpt = pack16(L, KT, 1, Ci, 4 * Co, 0, 1, Ci, 0)
    t = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    L.dc_convT2x2_fwd_f16x3(ad.data_ptr(), wpt.data_ptr(), None, t.data_ptr(), Co, None, None, None, 0, bd.data_ptr(), 0, None, 0,
                            N, H, W, Ci, Co, None)
    dzt = (rs.standard_normal((N, 2 * H, 2 * W, Co)) * 1e-6).astype(np.float32)
    wst = torch.empty(L.dc_convT2x2_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    g = torch.full((2, 2, Co, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_wgrad_f16x3(ad.data_ptr(), dev(dzt).data_ptr(), g.data_ptr(), wst.data_ptr(), scl.data_ptr(), bd.data_ptr(),
                              N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(t.cpu().numpy(), on.convT2x2_fwd(a.astype(np.float64), KT.astype(np.float64), np.zeros(Co))) < 2e-5
    _, gK_ref, _ = on.convT2x2_bwd(a.astype(np.float64), KT.astype(np.float64), dzt.astype(np.float64))
    assert torch.isfinite(g).all() and rel_err(g.cpu().numpy(), gK_ref) < 2e-5
    if amag > 65504:
        L.dc_conv3x3_fwd_f16x3(ad.data_ptr(), wp.data_ptr(), None, z.data_ptr(), Co, None, 0, None, None, 0, None, 0, None, 0, None,
                               N, H, W, Ci, Co, None)
        torch.cuda.synchronize()
        assert not torch.isfinite(z).all()


def test_bn_statistics_at_mean_1000_sigma(dclib):
    """|mean| / sigma = 1e3 per channel (a large conv bias): the (sum, sum of squares) partials are formed from shifted
    sums + Chan merges (csrc/common.h DcMoments), so mean and 1/sqrt(var + eps) match the float64 oracle; plain fp32
    sums of v and v*v lose the variance here.  Both contraction paths, interior and ragged tiles."""
    L = dclib
    N, H, W, Ci, Co = 2, 40, 40, 32, 64
    rs = np.random.RandomState(21)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(1.0 / (9 * Ci))).astype(np.float32)
    b = (np.where(rs.random_sample(Co) < 0.5, -1.0, 1.0) * 1000.0).astype(np.float32)
    z_ref = on.conv3x3_fwd(x.astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    mu, var = z_ref.mean((0, 1, 2)), z_ref.var((0, 1, 2))
    assert (np.abs(mu) / np.sqrt(var)).min() > 500
    M = N * H * W
    wp = pack16(L, K, 9, Ci, Co, Ci * Co, Co, 1, 0)
    for entry in ('f16x3', 'f32'):
        z = torch.empty((N, H, W, Co), device='cuda')
        tiles = L.dc_conv3x3_tiles(N, H, W, Co)
        stats = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
        if entry == 'f16x3':
            L.dc_conv3x3_fwd_f16x3(dev(x).data_ptr(), wp.data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(), 0,
                                   None, None, 0, None, 0, None, 0, None, N, H, W, Ci, Co, None)
        else:
            wp32 = torch.empty(9 * Ci * Co, device='cuda')
            L.dc_pack_weights(dev(K).data_ptr(), wp32.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
            L.dc_conv3x3_fwd(dev(x).data_ptr(), wp32.data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(), None, None,
                             0, N, H, W, Ci, Co, None)
        mean, invstd = torch.empty(Co, device='cuda'), torch.empty(Co, device='cuda')
        L.dc_bn_stats_finalize(stats.data_ptr(), tiles, 1, Co, float(M), 1e-3, -1.0, mean.data_ptr(), invstd.data_ptr(), None,
                               None, None)
        torch.cuda.synchronize()
        assert np.allclose(mean.cpu().numpy(), mu, rtol=1e-6), entry
        assert np.allclose(invstd.cpu().numpy(), 1 / np.sqrt(var + 1e-3), rtol=2e-4), entry


def test_first_layer_statistics_at_large_offset(dclib):
    L = dclib
    N, H, W, Co = 2, 64, 64, 32
    rs = np.random.RandomState(4)
    x = rs.standard_normal((N, H, W)).astype(np.float32)
    K = (rs.standard_normal((3, 3, 1, Co)) / 3).astype(np.float32)
    b = np.full(Co, 2000.0, np.float32)
    z_ref = on.conv3x3_fwd(x[..., None].astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    tiles = L.dc_conv3x3_c1_tiles(N, H, W, Co)
    stats = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
    z = torch.empty((N, H, W, Co), device='cuda')
    L.dc_conv3x3_c1_fwd(dev(x).data_ptr(), dev(K).data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(), None, None, 0,
                        None, 0, N, H, W, Co, None)
    mean, invstd = torch.empty(Co, device='cuda'), torch.empty(Co, device='cuda')
    L.dc_bn_stats_finalize(stats.data_ptr(), tiles, 1, Co, float(N * H * W), 1e-3, -1.0, mean.data_ptr(), invstd.data_ptr(), None,
                           None, None)
    torch.cuda.synchronize()
    assert np.allclose(invstd.cpu().numpy(), 1 / np.sqrt(z_ref.var((0, 1, 2)) + 1e-3), rtol=2e-4)


# ---- whole network -------------------------------------------------------------------------------------------------
def _scaled_weights(nfb, kscale, gscale, seed=31):
    """Keras-order weights with every conv / conv-transpose kernel multiplied by kscale and every gamma / beta by
    gscale (BatchNorm renormalises the kernel scale away up to eps; gamma scales every activation)."""
    Wt = on.init_weights(nfb, seed=seed, randomize_bn=True)
    out, i = [], 0
    for name, kind, cin, cout, mom in on.layer_table(nfb):
        n = 2 if kind == 'head' else 6
        ws = [np.array(w, np.float32) for w in Wt[i:i + n]]
        if kind != 'head':
            ws[0] = (ws[0] * kscale).astype(np.float32)
            ws[2] = (ws[2] * gscale).astype(np.float32)
            ws[3] = (ws[3] * gscale).astype(np.float32)
            ws[5] = (ws[5] * kscale * kscale).astype(np.float32)        # moving variance follows the kernel scale
            ws[4] = (ws[4] * kscale).astype(np.float32)
        else:
            ws[0] = (ws[0] / gscale).astype(np.float32)                  # keep the logits O(1)
        out += ws
        i += n
    return out


# per-tensor gradient bound (rel-L2 and max-abs / max) with the oracle's gates forced: 1e-3 under these scalings -- except
# kernels x 2^-12 alone: the pre-BN variance (~1e-8) sits far below BatchNorm's eps = 1e-3, xhat is ~1e-3 and dgamma = sum dy * xhat
# cancels to ~1e-2 of its terms, so fp32 rounding shows at the percent level in that one quantity (d0b's dgamma: 1.6e-2)
@pytest.mark.parametrize('kscale,gscale,GRAD_REL', [(2.0 ** -12, 1.0, 0.05), (2.0 ** 8, 1.0, 1e-3), (1.0, 3e4, 1e-3), (2.0 ** -10, 1e4, 1e-3)])
def test_train_step_with_extreme_weights_matches_oracle(kscale, gscale, GRAD_REL):
    """set_weights() with kernels x 2^-12 / x 2^8 and gamma, beta x 3e4 (activations ~1e5 > 65504): one training forward
    + backward against the float64 oracle -- probabilities and loss at 1e-4, gradients in rel-L2 -- and nothing is inf."""
    from deep_calcium_amd.net import UNetEngine
    N, H, W, nfb = 2, 32, 32, 8
    Wt = _scaled_weights(nfb, kscale, gscale)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    from _forced import assert_forcing_is_benign, device_decisions, grad_report
    eng = UNetEngine((H, W), nb_filters_base=nfb)
    eng.set_weights(Wt)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p = eng.forward_train(xd, yd, {k: torch.from_numpy(v).cuda() for k, v in masks.items()}, update_moving=False).cpu().numpy()
    loss = eng.read_sums()[0] / p.size
    eng.backward()
    G = eng.grads()
    # the oracle through the device's own ReLU gates / pool indices (tests/_forced.py): what is compared is rounding
    loss_ref, p_ref, G_ref, _ = on.UNetOracle(Wt, nfb, force=device_decisions(eng, N)).loss_and_grads(x, y, masks)
    assert np.isfinite(p).all() and np.isfinite(eng.gflat.cpu().numpy()).all()
    if gscale > 1:
        assert max(float(eng.activation(n, N).abs().max()) for n in ('e0b', 'e1a', 'd1a')) > 65504      # really beyond fp16
    assert np.abs(p - p_ref).max() < 1e-4 and abs(loss - loss_ref) < 1e-4
    assert_forcing_is_benign(Wt, nfb, x, masks, p_ref, p_dev=p)
    keep = lambda n, j: not (j == 1 and n != 'out')          # conv biases in front of BN: analytically zero gradient
    fg = np.concatenate([g.ravel() for n in G_ref for j, g in enumerate(G[n]) if keep(n, j)]).astype(np.float64)
    fr = np.concatenate([g.ravel() for n in G_ref for j, g in enumerate(G_ref[n]) if keep(n, j)])
    # per-tensor comparison (the kernel gradients differ by orders of magnitude between layers under these scalings)
    for n in G_ref:
        for j, g in enumerate(G_ref[n]):
            if keep(n, j) and np.linalg.norm(g) > 0:
                rel = np.linalg.norm(G[n][j].astype(np.float64) - g) / np.linalg.norm(g)
                assert rel < GRAD_REL, (n, j, rel)
    worst, rel_all, _ = grad_report(G, G_ref, 'extreme weights k x %g, gamma x %g, forced gates: ' % (kscale, gscale))
    assert worst < GRAD_REL, worst
    assert fg.dot(fr) / (np.linalg.norm(fg) * np.linalg.norm(fr)) > (0.999 if GRAD_REL > 1e-3 else 0.999999)


@pytest.mark.parametrize('xscale', [1e4, 1e-4])
def test_inference_on_unnormalised_input(xscale, monkeypatch):
    """predict() on images that were NOT normalised (a custom series_summary_func): folded-BN activations of ~1e5 go
    through the measured-max range guard.  At that magnitude the logits are ~1e5 and fp32 itself resolves them to ~1e-2,
    so the yardstick is relative: every tapped activation within 2e-5 of the oracle's scale, the probabilities as close
    to the oracle as the library's own fp32-MFMA path (no fp16 anywhere) gets -- and no inf / nan."""
    from deep_calcium_amd.net import UNetEngine
    monkeypatch.delenv('DC_INFER_GUARD', raising=False)        # starts in the (default) optimistic mode
    N, H, W, nfb = 2, 48, 48, 8
    Wt = on.init_weights(nfb, seed=5, randomize_bn=True)
    x, _ = on.synthetic_batch(N, H, W)
    x = (x * xscale).astype(np.float32)
    taps = {}
    p_ref = on.UNetOracle(Wt, nfb).forward(x, training=False, taps=taps)
    res = {}
    for mode in ('f16x3', 'f32'):
        eng = UNetEngine((H, W), nb_filters_base=nfb, mfma=mode)
        eng.set_weights(Wt)
        xd = torch.from_numpy(x).cuda()
        if mode == 'f16x3' and xscale > 1:
            # the optimistic pass (no activation scale) overflows fp16 and says so ...
            assert not eng.infer_measured
            eng.forward_infer(xd)
            torch.cuda.synchronize()
            assert float(eng._ovf[0].item()) == 1.0
        # ... and the checked entry point (what Model.predict / predict_tta call) repeats it with measured bounds
        p = eng.forward_infer_checked(xd).cpu().numpy()
        assert eng.infer_measured == (mode == 'f16x3' and xscale > 1)
        A = eng._acts(N)
        assert np.isfinite(p).all()
        errs = {}
        for name in ('bb', 'd3b', 'd1b', 'd0b'):
            got = A[name].cpu().numpy()
            errs[name] = np.abs(got - taps[name]).max() / np.abs(taps[name]).max()
        res[mode] = (np.abs(p - p_ref).max(), errs)
        if xscale > 1 and mode == 'f16x3':
            assert float(A['e0a'].abs().max()) > 65504                  # really beyond fp16's range
    for name, e in res['f16x3'][1].items():
        assert e < 2e-5, (name, e, res['f32'][1][name])
    assert res['f16x3'][0] < max(1e-4, 3 * res['f32'][0]), res
