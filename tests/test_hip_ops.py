"""GPU parity of every libdcunet entry point against the float64 numpy oracle (called through the C ABI).

Tolerances: fp32 arithmetic vs a float64 oracle -> relative 2e-5 of the output scale for the MFMA
contractions (k-ordered fp32 fma chains), bit-exact for pool argmax indices and uint8 outputs.
"""
import numpy as np
import pytest

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu


def _fused_paths_enabled(*knobs):
    """The A/B knobs that switch a fused entry point off make its `*_blocks()` query return 0: skip, don't fail."""
    import os
    return all(os.environ.get(k, '1') != '0' for k in ('DC_IGEMM_PP',) + knobs)

torch = pytest.importorskip('torch')


_KEEP = []


@pytest.fixture(autouse=True)
def _keep_alive():
    """dev() temporaries must outlive the asynchronous launches that read them (raw pointers cross the C ABI)."""
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    t = t.cuda()
    _KEEP.append(t)
    return t


def rel_err(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


def pack(L, K, taps, Kdim, Ncols, s_tap, s_k, s_n, flip):
    src = dev(K.astype(np.float32))
    dst = torch.empty(taps * Kdim * Ncols, dtype=torch.float32, device='cuda')
    L.dc_pack_weights(src.data_ptr(), dst.data_ptr(), taps, Kdim, Ncols, s_tap, s_k, s_n, flip, None)
    return dst


CONV_SHAPES = [
    # N, H, W, Cin, Cout          tile config exercised
    (2, 32, 32, 32, 64),        # A32
    (1, 64, 64, 64, 32),        # B32 (Cout <= 32)
    (2, 16, 16, 128, 128),      # A16
    (2, 8, 8, 64, 256),         # C8
    (1, 20, 36, 8, 24),         # ragged: partial tiles, Cin < CK, Cout not a multiple of 32
    (3, 40, 40, 32, 32),        # B32, partial tiles in y
    (1, 6, 6, 16, 64),          # 96^2-window bottleneck size
    (1, 32, 64, 128, 64),       # deep-layer shape at W > 16: many chunks, interior tiles
    (2, 40, 40, 144, 96),       # ragged in y, x and columns, Cout % 64 != 0
    (1, 24, 48, 136, 64),       # partial last chunk (Cin % 16 = 8)
    (20, 16, 16, 256, 256),     # the 16^2 layers of the reference's 128^2 x 20 training step: 4-way split-K + combine launch
    (20, 8, 8, 512, 512),       # its 8^2 layers: split-K on the 8 x 8 tiles
    (3, 12, 12, 200, 96),       # split-K with a partial tile, a partial last chunk and Cout % 64 != 0
]


@pytest.mark.parametrize('N,H,W,Ci,Co', CONV_SHAPES)
def test_conv3x3_fwd_dgrad_wgrad(dclib, N, H, W, Ci, Co):
    L = dclib
    rs = np.random.RandomState(N * 1000 + H + Ci)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    dz = rs.standard_normal((N, H, W, Co)).astype(np.float32)
    z_ref = on.conv3x3_fwd(x.astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    dx_ref, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))

    xd, bd, dzd = dev(x), dev(b), dev(dz)
    wp = pack(L, K, 9, Ci, Co, Ci * Co, Co, 1, 0)
    wpd = pack(L, K, 9, Co, Ci, Ci * Co, 1, Co, 1)
    z = torch.full((N, H, W, Co), float('nan'), device='cuda')
    tiles = L.dc_conv3x3_tiles(N, H, W, Co)
    stats = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
    L.dc_conv3x3_fwd(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), z.data_ptr(), Co, stats.data_ptr(), None, None, 0,
                     N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z.cpu().numpy(), z_ref) < 2e-5
    st = stats.cpu().numpy().reshape(tiles, Co, 2).astype(np.float64).sum(0)
    assert np.allclose(st[:, 0], z_ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-3 * np.sqrt(N * H * W))
    assert np.allclose(st[:, 1], (z_ref ** 2).sum((0, 1, 2)), rtol=1e-4)

    # fused inference epilogue into a strided (concat) destination
    sc = rs.uniform(0.5, 1.5, Co).astype(np.float32)
    sh = rs.uniform(-0.5, 0.5, Co).astype(np.float32)
    cat = torch.zeros((N, H, W, 2 * Co), device='cuda')
    L.dc_conv3x3_fwd(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), cat.data_ptr() + 4 * Co, 2 * Co, None,
                     dev(sc).data_ptr(), dev(sh).data_ptr(), 1, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    c = cat.cpu().numpy()
    assert np.all(c[..., :Co] == 0)
    assert rel_err(c[..., Co:], np.maximum(z_ref * sc + sh, 0)) < 2e-5

    dx = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad(dzd.data_ptr(), wpd.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    dw = torch.full((3, 3, Ci, Co), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad(xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dx_ref) < 2e-5
    assert rel_err(dw.cpu().numpy(), dK_ref) < 2e-5


@pytest.mark.parametrize('N,H,W,Ci,Co', CONV_SHAPES)
def test_conv3x3_f16x3_fwd_dgrad(dclib, N, H, W, Ci, Co):
    """Split-fp16 contraction: fp32-grade accuracy (tolerance as tight as the fp32-MFMA path), gradient-sized
    inputs (1e-7) survive thanks to the exact power-of-two input scale."""
    L = dclib
    rs = np.random.RandomState(N * 1000 + H + Ci)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    dz = (rs.standard_normal((N, H, W, Co)) * 3e-7).astype(np.float32)          # gradient magnitudes (1/(N*H*W))
    z_ref = on.conv3x3_fwd(x.astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    dx_ref, _, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    Kd = dev(K)
    wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Co, Ci), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Co, Ci, Ci * Co, 1, Co, 1, None)
    z = torch.full((N, H, W, Co), float('nan'), device='cuda')
    tiles = L.dc_conv3x3_tiles(N, H, W, Co)
    stats = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
    # the caller owns the split-K slabs (dcunet.h): sized by the query, 0 floats = this shape never splits
    skf, skd = L.dc_conv3x3_splitk_ws_floats(N, H, W, Ci, Co, 0), L.dc_conv3x3_splitk_ws_floats(N, H, W, Ci, Co, 1)
    if (N, H, W, Ci, Co) in ((20, 16, 16, 256, 256), (20, 8, 8, 512, 512), (3, 12, 12, 200, 96)):
        assert skf > 0 and skf % (N * H * W * Co) == 0 and 2 <= skf // (N * H * W * Co) <= 4
    if W > 16:
        assert skf == 0 and skd == 0
    skws = torch.full((max(skf, skd, 4),), float('nan'), device='cuda')
    L.dc_conv3x3_fwd_f16x3(dev(x).data_ptr(), wp.data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(), 0,
                           None, None, 0, None, 0, None, 0, skws.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z.cpu().numpy(), z_ref) < 2e-5
    st = stats.cpu().numpy().reshape(tiles, Co, 2).astype(np.float64).sum(0)
    assert np.allclose(st[:, 0], z_ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-3 * np.sqrt(N * H * W))
    assert np.allclose(st[:, 1], (z_ref ** 2).sum((0, 1, 2)), rtol=1e-4)
    if skf > 0:      # no workspace: ONE un-split launch -- same values up to the summation order, same statistics rows
        z1 = torch.full((N, H, W, Co), float('nan'), device='cuda')
        stats1 = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
        L.dc_conv3x3_fwd_f16x3(dev(x).data_ptr(), wp.data_ptr(), dev(b).data_ptr(), z1.data_ptr(), Co, stats1.data_ptr(), 0,
                               None, None, 0, None, 0, None, 0, None, N, H, W, Ci, Co, None)
        torch.cuda.synchronize()
        assert rel_err(z1.cpu().numpy(), z_ref) < 2e-5 and not torch.equal(z, z1)
        assert np.allclose(stats1.cpu().numpy(), stats.cpu().numpy(), rtol=1e-4, atol=1e-3)
    amax = dev(np.array([np.abs(dz).max()], np.float32))
    scl = torch.empty(1, device='cuda')
    L.dc_pow2_scale_from_absmax(amax.data_ptr(), 1, 1024.0, scl.data_ptr(), None)
    dx = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_f16x3(dev(dz).data_ptr(), wpd.data_ptr(), dx.data_ptr(), scl.data_ptr(), None, 0, skws.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dx_ref) < 2e-5
    # the scale derived IN the kernel from per-block maxima (what dc_bn_bwd_apply leaves behind; the true maximum sits in
    # one of 37 entries, the others are smaller): same power of two, same bits
    part = np.full(37, np.abs(dz).max() * 0.3, np.float32)
    part[11] = np.abs(dz).max()
    dxa = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_f16x3(dev(dz).data_ptr(), wpd.data_ptr(), dxa.data_ptr(), None, dev(part).data_ptr(), 37, skws.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(dx, dxa)
    with pytest.raises(Exception, match='not both'):
        L.dc_conv3x3_dgrad_f16x3(dev(dz).data_ptr(), wpd.data_ptr(), dxa.data_ptr(), scl.data_ptr(), dev(part).data_ptr(), 37, None, N, H, W, Ci, Co, None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    dw = torch.full((3, 3, Ci, Co), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_f16x3(dev(x).data_ptr(), dev(dz).data_ptr(), dw.data_ptr(), ws.data_ptr(), scl.data_ptr(),
                             None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    _, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    assert rel_err(dw.cpu().numpy(), dK_ref) < 2e-5
    # without the scale the same input underflows fp16 and the result is garbage-level: the scale is load-bearing
    L.dc_conv3x3_dgrad_f16x3(dev(dz).data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, None, 0, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu().numpy(), dx_ref) > 1e-3


@pytest.mark.parametrize('N,H,W,Ci,Co', [(2, 32, 32, 32, 64), (1, 64, 64, 64, 32), (2, 16, 16, 128, 128),
                                         (1, 20, 36, 8, 24), (3, 40, 40, 32, 32), (1, 6, 6, 16, 64), (1, 24, 40, 128, 64)])
def test_bn_relu_on_load_conv_convT_wgrad_head(dclib, N, H, W, Ci, Co):
    """"bnin" entry points: the consumer takes the producer's pre-BN tensor z plus the per-channel training affine
    and forms relu(fmaf(z, sc, sh)) while staging -- results must equal the materialised path (and the oracle),
    zero padding must stay zero (a shift > 0 would otherwise leak relu(sh) into the border)."""
    L = dclib
    rs = np.random.RandomState(N * 77 + H + Ci)
    zin = rs.standard_normal((N, H, W, Ci)).astype(np.float32) * 2 + 0.5
    gamma = (rs.random_sample(Ci) + 0.5).astype(np.float32)
    beta = (rs.standard_normal(Ci) * 0.5 + 0.3).astype(np.float32)           # mostly positive shifts
    # statistics through the library: partial sums -> finalize_affine (mean, invstd, scale, shift)
    part = np.stack([zin.reshape(-1, Ci).astype(np.float64).sum(0), (zin.reshape(-1, Ci).astype(np.float64) ** 2).sum(0)], -1)
    mean, invstd, sc, sh = (torch.empty(Ci, device='cuda') for _ in range(4))
    ab = torch.empty(Ci, device='cuda')
    L.dc_bn_stats_finalize_affine(dev(part).data_ptr(), 1, 1, Ci, float(N * H * W), 1e-3, -1.0,
                                  mean.data_ptr(), invstd.data_ptr(), None, None, dev(gamma).data_ptr(),
                                  dev(beta).data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), None)
    # the materialised activation from the library's own BN kernel (same affine, same bits)
    a = torch.empty(N, H, W, Ci, device='cuda')
    L.dc_bn_relu_drop_fwd(dev(zin).data_ptr(), mean.data_ptr(), invstd.data_ptr(), dev(gamma).data_ptr(),
                          dev(beta).data_ptr(), None, 1.0, 0, a.data_ptr(), Ci, N * H * W, Ci, 0.0, None, None)
    torch.cuda.synchronize()
    a_np = a.cpu().numpy()
    # the range-guard bound really bounds the activation, per channel
    assert np.allclose(ab.cpu().numpy(), np.abs(gamma) * np.sqrt(N * H * W) + np.abs(beta), rtol=1e-6)
    assert (a_np.reshape(-1, Ci).max(0) <= ab.cpu().numpy()).all()
    mu64 = zin.reshape(-1, Ci).astype(np.float64).mean(0)
    var64 = zin.reshape(-1, Ci).astype(np.float64).var(0)
    a_ref = np.maximum((zin.astype(np.float64) - mu64) / np.sqrt(var64 + 1e-3) * gamma + beta, 0)
    assert np.abs(a_np - a_ref).max() < 1e-5 * max(1.0, np.abs(a_ref).max())
    assert (a_np > 0).mean() > 0.3

    # conv3x3 forward
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    Kd = dev(K)
    wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    z1 = torch.full((N, H, W, Co), float('nan'), device='cuda'); z2 = torch.full_like(z1, float('nan'))
    L.dc_conv3x3_fwd_f16x3(a.data_ptr(), wp.data_ptr(), None, z1.data_ptr(), Co, None, 0, None, None, 0, ab.data_ptr(), 0, None, 0, None,
                           N, H, W, Ci, Co, None)
    L.dc_conv3x3_fwd_bnin_f16x3(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), wp.data_ptr(), None,
                                z2.data_ptr(), Co, None, 0, None, None, 0, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(z1, z2)                       # same operand bits -> same result bits
    z_ref = on.conv3x3_fwd(a_ref, K.astype(np.float64), np.zeros(Co))
    assert rel_err(z2.cpu().numpy(), z_ref) < 2e-5

    # conv3x3 weight gradient
    dz = (rs.standard_normal((N, H, W, Co)) * 3e-7).astype(np.float32)
    scl = torch.empty(1, device='cuda')
    L.dc_pow2_scale_from_absmax(dev(np.array([np.abs(dz).max()], np.float32)).data_ptr(), 1, 1024.0, scl.data_ptr(), None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    dw1 = torch.full((3, 3, Ci, Co), float('nan'), device='cuda'); dw2 = torch.full_like(dw1, float('nan'))
    dzd = dev(dz)
    L.dc_conv3x3_wgrad_f16x3(a.data_ptr(), dzd.data_ptr(), dw1.data_ptr(), ws.data_ptr(), scl.data_ptr(), ab.data_ptr(),
                             N, H, W, Ci, Co, None)
    L.dc_conv3x3_wgrad_bnin_f16x3(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), dzd.data_ptr(),
                                  dw2.data_ptr(), ws.data_ptr(), scl.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2)
    _, dK_ref, _ = on.conv3x3_bwd(a_ref, K.astype(np.float64), dz.astype(np.float64))
    assert rel_err(dw2.cpu().numpy(), dK_ref) < 2e-5

    # conv-transpose forward + weight gradient (x is the B operand there)
    KT = (rs.standard_normal((2, 2, Co, Ci)) * np.sqrt(2.0 / (4 * Ci))).astype(np.float32)     # Keras (kh,kw,out,in)
    KTd = dev(KT)
    wpt = torch.empty(L.dc_pack_weights_f16x3_floats(1, Ci, 4 * Co), device='cuda')
    L.dc_pack_weights_f16x3(KTd.data_ptr(), wpt.data_ptr(), 1, Ci, 4 * Co, 0, 1, Ci, 0, None)
    t1 = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device='cuda'); t2 = torch.full_like(t1, float('nan'))
    L.dc_convT2x2_fwd_f16x3(a.data_ptr(), wpt.data_ptr(), None, t1.data_ptr(), Co, None, None, None, 0, ab.data_ptr(), 0, None, 0,
                            N, H, W, Ci, Co, None)
    L.dc_convT2x2_fwd_bnin_f16x3(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), wpt.data_ptr(), None,
                                 t2.data_ptr(), Co, None, None, None, 0, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(t1, t2)
    assert rel_err(t2.cpu().numpy(), on.convT2x2_fwd(a_ref, KT.astype(np.float64), np.zeros(Co))) < 2e-5
    dzt = (rs.standard_normal((N, 2 * H, 2 * W, Co)) * 3e-7).astype(np.float32)
    wst = torch.empty(L.dc_convT2x2_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    g1 = torch.full((2, 2, Co, Ci), float('nan'), device='cuda'); g2 = torch.full_like(g1, float('nan'))
    dztd = dev(dzt)
    L.dc_convT2x2_wgrad_f16x3(a.data_ptr(), dztd.data_ptr(), g1.data_ptr(), wst.data_ptr(), scl.data_ptr(), ab.data_ptr(),
                              N, H, W, Ci, Co, None)
    L.dc_convT2x2_wgrad_bnin_f16x3(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), ab.data_ptr(), dztd.data_ptr(),
                                   g2.data_ptr(), wst.data_ptr(), scl.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(g1, g2)

    # head (needs a power-of-two channel count)
    if Ci & (Ci - 1) == 0:
        pixels = N * H * W
        kh = (rs.standard_normal((Ci, 2)) * 0.3).astype(np.float32); bh = rs.standard_normal(2).astype(np.float32)
        y = (rs.random_sample(pixels) < 0.3).astype(np.uint8)
        hb = L.dc_head_blocks(pixels)
        p1, p2 = torch.empty(pixels, device='cuda'), torch.empty(pixels, device='cuda')
        q1, q2 = torch.zeros(hb * 12, device='cuda'), torch.zeros(hb * 12, device='cuda')
        L.dc_head_fwd(a.data_ptr(), dev(kh).data_ptr(), dev(bh).data_ptr(), dev(y).data_ptr(), p1.data_ptr(), q1.data_ptr(), pixels, Ci, None)
        L.dc_head_fwd_bnin(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), dev(kh).data_ptr(), dev(bh).data_ptr(),
                           dev(y).data_ptr(), p2.data_ptr(), q2.data_ptr(), pixels, Ci, None)
        torch.cuda.synchronize()
        assert torch.equal(p1, p2) and torch.equal(q1, q2)
        d1, d2 = torch.empty(pixels, Ci, device='cuda'), torch.empty(pixels, Ci, device='cuda')
        r1, r2 = torch.zeros(hb * (Ci + 4), device='cuda'), torch.zeros(hb * (Ci + 4), device='cuda')
        L.dc_head_bwd(a.data_ptr(), p1.data_ptr(), dev(y).data_ptr(), dev(kh).data_ptr(), d1.data_ptr(), r1.data_ptr(), 0, None, pixels, Ci, None)
        L.dc_head_bwd_bnin(dev(zin).data_ptr(), sc.data_ptr(), sh.data_ptr(), p1.data_ptr(), dev(y).data_ptr(),
                           dev(kh).data_ptr(), d2.data_ptr(), r2.data_ptr(), 0, None, pixels, Ci, None)
        torch.cuda.synchronize()
        assert torch.equal(d1, d2) and torch.equal(r1, r2)


def test_conv3x3_identity_kernel_kat(dclib):
    """Analytic known-answer: centre-tap identity kernel reproduces the input exactly; a one-hot shifted
    tap reproduces the zero-padded shift (pins 'same' padding + cross-correlation orientation)."""
    L = dclib
    N, H, W, C = 1, 32, 32, 32
    rs = np.random.RandomState(3)
    x = rs.standard_normal((N, H, W, C)).astype(np.float32)
    for (a, b) in ((1, 1), (0, 2), (2, 0)):
        K = np.zeros((3, 3, C, C), np.float32)
        K[a, b] = np.eye(C)
        wp = pack(L, K, 9, C, C, C * C, C, 1, 0)
        z = torch.empty((N, H, W, C), device='cuda')
        L.dc_conv3x3_fwd(dev(x).data_ptr(), wp.data_ptr(), None, z.data_ptr(), C, None, None, None, 0, N, H, W, C, C, None)
        torch.cuda.synchronize()
        exp = np.zeros_like(x)
        xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        exp[:] = xp[:, a:a + H, b:b + W]
        assert np.array_equal(z.cpu().numpy(), exp)


@pytest.mark.parametrize('N,H,W,Co', [(2, 32, 32, 32), (1, 20, 36, 8), (3, 16, 16, 4), (1, 6, 10, 16), (2, 64, 64, 32)])
def test_conv3x3_c1(dclib, N, H, W, Co):
    L = dclib
    rs = np.random.RandomState(5)
    x = rs.standard_normal((N, H, W)).astype(np.float32)
    K = rs.standard_normal((3, 3, 1, Co)).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    dz = rs.standard_normal((N, H, W, Co)).astype(np.float32)
    z_ref = on.conv3x3_fwd(x[..., None].astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    _, dK_ref, _ = on.conv3x3_bwd(x[..., None].astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    tiles = L.dc_conv3x3_c1_tiles(N, H, W, Co)
    stats = torch.zeros(tiles * Co * 2, device='cuda', dtype=torch.float64)
    z = torch.empty((N, H, W, Co), device='cuda')
    amx = torch.zeros(Co, device='cuda')
    L.dc_conv3x3_c1_fwd(dev(x).data_ptr(), dev(K).data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(),
                        None, None, 0, amx.data_ptr(), 0, N, H, W, Co, None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, 1, Co), device='cuda')
    dw = torch.empty((3, 3, 1, Co), device='cuda')
    L.dc_conv3x3_wgrad(dev(x).data_ptr(), dev(dz).data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, 1, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z.cpu().numpy(), z_ref) < 1e-5
    st = stats.cpu().numpy().reshape(tiles, Co, 2).astype(np.float64).sum(0)
    assert np.allclose(st[:, 0], z_ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
    assert np.allclose(st[:, 1], (z_ref ** 2).sum((0, 1, 2)), rtol=1e-5)
    assert np.allclose(amx.cpu().numpy(), np.abs(z.cpu().numpy()).max((0, 1, 2)), rtol=0, atol=0)     # measured max |out|
    assert rel_err(dw.cpu().numpy(), dK_ref) < 1e-5


CONVT_SHAPES = [(2, 32, 32, 64, 32), (1, 16, 16, 128, 64), (2, 8, 8, 256, 128), (1, 10, 18, 16, 8), (1, 4, 4, 512, 256),
                (1, 34, 40, 64, 64), (2, 33, 64, 32, 96)]       # W > 16: 256-column / 128-column workgroups on 2- / 4-row tiles, ragged


@pytest.mark.parametrize('N,H,W,Ci,Co', CONVT_SHAPES)
def test_convT2x2(dclib, N, H, W, Ci, Co):
    L = dclib
    rs = np.random.RandomState(11 + H)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((2, 2, Co, Ci)) * np.sqrt(2.0 / (4 * Co))).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    dz = rs.standard_normal((N, 2 * H, 2 * W, Co)).astype(np.float32)
    z_ref = on.convT2x2_fwd(x.astype(np.float64), K.astype(np.float64), b.astype(np.float64))
    dx_ref, dK_ref, _ = on.convT2x2_bwd(x.astype(np.float64), K.astype(np.float64), dz.astype(np.float64))
    wp = pack(L, K, 1, Ci, 4 * Co, 0, 1, Ci, 0)
    wpd = pack(L, K, 4, Co, Ci, Co * Ci, Ci, 1, 0)
    tiles = L.dc_convT2x2_tiles(N, H, W, Co)
    stats = torch.zeros(tiles * 4 * Co * 2, device='cuda', dtype=torch.float64)
    z = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    L.dc_convT2x2_fwd(dev(x).data_ptr(), wp.data_ptr(), dev(b).data_ptr(), z.data_ptr(), Co, stats.data_ptr(),
                      None, None, 0, N, H, W, Ci, Co, None)
    dx = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_dgrad(dev(dz).data_ptr(), wpd.data_ptr(), dx.data_ptr(), N, H, W, Ci, Co, None)
    ws = torch.empty(L.dc_convT2x2_wgrad_ws_floats(N, H, W, Ci, Co), device='cuda')
    dw = torch.full((2, 2, Co, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_wgrad(dev(x).data_ptr(), dev(dz).data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z.cpu().numpy(), z_ref) < 2e-5
    st = stats.cpu().numpy().reshape(tiles, 4, Co, 2).astype(np.float64).sum((0, 1))
    assert np.allclose(st[:, 0], z_ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
    assert np.allclose(st[:, 1], (z_ref ** 2).sum((0, 1, 2)), rtol=1e-4)
    assert rel_err(dx.cpu().numpy(), dx_ref) < 2e-5
    assert rel_err(dw.cpu().numpy(), dK_ref) < 2e-5
    # split-fp16 variants on gradient-sized dz (scaled exactly by a power of two)
    dzs = (dz * 1e-7).astype(np.float32)
    scl = torch.empty(1, device='cuda')
    L.dc_pow2_scale_from_absmax(dev(np.array([np.abs(dzs).max()], np.float32)).data_ptr(), 1, 1024.0, scl.data_ptr(), None)
    Kd = dev(K)
    wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(1, Ci, 4 * Co), device='cuda')
    wpd16 = torch.empty(L.dc_pack_weights_f16x3_floats(4, Co, Ci), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wp16.data_ptr(), 1, Ci, 4 * Co, 0, 1, Ci, 0, None)
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd16.data_ptr(), 4, Co, Ci, Co * Ci, Ci, 1, 0, None)
    z2 = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    tiles16 = L.dc_convT2x2_f16x3_tiles(N, H, W, Co)          # the split-fp16 forward runs wider column blocks on flatter tiles
    stats2 = torch.full((tiles16 * 4 * Co * 2,), float('nan'), device='cuda', dtype=torch.float64)
    amx = torch.zeros(Co, device='cuda')
    L.dc_convT2x2_fwd_f16x3(dev(x).data_ptr(), wp16.data_ptr(), dev(b).data_ptr(), z2.data_ptr(), Co, None,
                            None, None, 0, None, 0, amx.data_ptr(), 0, N, H, W, Ci, Co, None)       # inference: measured max
    torch.cuda.synchronize()
    zi = z2.clone()
    L.dc_convT2x2_fwd_f16x3(dev(x).data_ptr(), wp16.data_ptr(), dev(b).data_ptr(), z2.data_ptr(), Co, stats2.data_ptr(),
                            None, None, 0, None, 0, None, 0, N, H, W, Ci, Co, None)                  # training: partials
    torch.cuda.synchronize()
    assert torch.equal(zi, z2)
    dx2 = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_dgrad_f16x3(dev(dzs).data_ptr(), wpd16.data_ptr(), dx2.data_ptr(), scl.data_ptr(), None, 0, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(z2.cpu().numpy(), z_ref) < 2e-5
    assert np.array_equal(amx.cpu().numpy(), np.abs(z2.cpu().numpy()).max((0, 1, 2)))
    st2 = stats2.cpu().numpy().reshape(tiles16, 4, Co, 2).astype(np.float64).sum((0, 1))
    assert np.isfinite(st2).all()                              # every row of the buffer was written
    assert np.allclose(st2[:, 1], (z_ref ** 2).sum((0, 1, 2)), rtol=1e-4)
    assert rel_err(dx2.cpu().numpy(), dx_ref * 1e-7) < 2e-5
    part = np.full(5, np.abs(dzs).max() * 0.5, np.float32)
    part[4] = np.abs(dzs).max()
    dx3 = torch.full((N, H, W, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_dgrad_f16x3(dev(dzs).data_ptr(), wpd16.data_ptr(), dx3.data_ptr(), None, dev(part).data_ptr(), 5, None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx3)                     # scale derived in the kernel from per-block maxima: same bits
    dw2 = torch.full((2, 2, Co, Ci), float('nan'), device='cuda')
    L.dc_convT2x2_wgrad_f16x3(dev(x).data_ptr(), dev(dzs).data_ptr(), dw2.data_ptr(), ws.data_ptr(), scl.data_ptr(),
                              None, N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert rel_err(dw2.cpu().numpy(), dK_ref * 1e-7) < 2e-5


def test_convT_onehot_is_pixel_replication(dclib):
    """KAT: K[a,b,o,c] = delta(o,c) for every (a,b) => each input pixel is replicated into its 2x2 block."""
    L = dclib
    N, H, W, C = 1, 16, 16, 32
    x = np.random.RandomState(2).standard_normal((N, H, W, C)).astype(np.float32)
    K = np.zeros((2, 2, C, C), np.float32)
    K[:, :] = np.eye(C)
    wp = pack(L, K, 1, C, 4 * C, 0, 1, C, 0)
    z = torch.empty((N, 2 * H, 2 * W, C), device='cuda')
    L.dc_convT2x2_fwd(dev(x).data_ptr(), wp.data_ptr(), None, z.data_ptr(), C, None, None, None, 0, N, H, W, C, C, None)
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy(), np.repeat(np.repeat(x, 2, 1), 2, 2))


@pytest.mark.parametrize('pixels_shape,C,keep', [((2, 16, 16), 32, 0.75), ((1, 24, 40), 8, 1.0), ((3, 8, 8), 256, 0.5)])
def test_batchnorm_relu_dropout_fwd_bwd(dclib, pixels_shape, C, keep):
    L = dclib
    rs = np.random.RandomState(C)
    N, H, W = pixels_shape
    M = N * H * W
    z = (rs.standard_normal((N, H, W, C)) * 1.7 + 0.4).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rs.uniform(-0.5, 0.5, C).astype(np.float32)
    da = rs.standard_normal((N, H, W, C)).astype(np.float32)
    mask = (rs.random_sample((N, H, W, C)) < keep).astype(np.uint8)
    z64 = z.astype(np.float64)
    y_ref, cache = on.bn_train_fwd(z64, gamma.astype(np.float64), beta.astype(np.float64))
    a_ref = np.maximum(y_ref, 0) * (mask / keep if keep < 1 else 1.0)
    dy = da.astype(np.float64) * (mask / keep if keep < 1 else 1.0) * (y_ref > 0)
    dz_ref, dg_ref, db_ref = on.bn_train_bwd(dy, gamma.astype(np.float64), cache)

    # statistics through the finalize kernel (one "tile" per image row block -> exercises the partial reduction)
    parts = N * H
    part = np.stack([z64.reshape(parts, W, C).sum(1), (z64 ** 2).reshape(parts, W, C).sum(1)], axis=-1)
    mean, invstd = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    mm, mv = dev(np.full(C, 0.25, np.float32)), dev(np.full(C, 2.0, np.float32))
    L.dc_bn_stats_finalize(dev(part).data_ptr(), parts, 1, C, float(M), 1e-3, 0.9, mean.data_ptr(), invstd.data_ptr(),
                           mm.data_ptr(), mv.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.allclose(mean.cpu().numpy(), cache[2], atol=2e-6)
    assert np.allclose(invstd.cpu().numpy(), cache[1], rtol=2e-5)
    assert np.allclose(mm.cpu().numpy(), 0.25 * 0.9 + cache[2] * 0.1, atol=1e-6)
    assert np.allclose(mv.cpu().numpy(), 2.0 * 0.9 + cache[3] * 0.1, rtol=1e-5)      # biased variance

    zd, gd, bd, dad, md = dev(z), dev(gamma), dev(beta), dev(da), dev(mask)
    mptr = md.data_ptr() if keep < 1 else None
    out = torch.zeros((N, H, W, 2 * C), device='cuda')          # strided destination (concat slice)
    L.dc_bn_relu_drop_fwd(zd.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), mptr, keep, 0,
                          out.data_ptr() + 4 * C, 2 * C, M, C, 0.0, None, None)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.all(o[..., :C] == 0)
    assert np.abs(o[..., C:] - a_ref).max() < 2e-5

    blocks = L.dc_bn_bwd_blocks(M, C)
    part1 = torch.empty(blocks * C * 2, device='cuda')
    part2 = torch.empty(blocks * C, device='cuda')
    dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    dz = torch.empty((N, H, W, C), device='cuda')
    dbias = torch.empty(C, device='cuda')
    tmp = torch.empty(32 * C, device='cuda')
    args = (dad.data_ptr(), C, zd.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), mptr, keep, 0)
    L.dc_bn_bwd_reduce(*args, part1.data_ptr(), None, M, C, None)
    L.dc_bn_bwd_finalize(part1.data_ptr(), blocks, C, dg.data_ptr(), db.data_ptr(), None)
    amax = torch.empty(blocks, device='cuda')
    scl = torch.empty(1, device='cuda')
    L.dc_bn_bwd_apply(*args, dg.data_ptr(), db.data_ptr(), dz.data_ptr(), part2.data_ptr(), amax.data_ptr(), M, C, None)
    L.dc_pow2_scale_from_absmax(amax.data_ptr(), blocks, 1024.0, scl.data_ptr(), None)
    L.dc_reduce_partials(part2.data_ptr(), blocks, C, 1.0, dbias.data_ptr(), tmp.data_ptr(), None)
    torch.cuda.synchronize()
    scale = np.abs(dg_ref).max()
    assert np.abs(dg.cpu().numpy() - dg_ref).max() < 2e-5 * max(scale, 1)
    assert np.abs(db.cpu().numpy() - db_ref).max() < 2e-5 * max(np.abs(db_ref).max(), 1)
    assert np.abs(dz.cpu().numpy() - dz_ref).max() < 3e-5 * max(np.abs(dz_ref).max(), 1)
    assert np.abs(dbias.cpu().numpy()).max() < 1e-2          # analytically zero
    mx = np.abs(dz.cpu().numpy()).max()
    assert amax.cpu().numpy().max() == mx
    sv = float(scl.cpu().numpy()[0])
    assert sv == 2.0 ** np.floor(np.log2(1024.0 / mx)) and 512 < sv * mx <= 1024


def test_bn_constant_input_kat(dclib):
    """KAT: constant input => var = 0 => relu(bn(z)) = relu(beta).  BN is applied in tf.nn.batch_normalization's own
    form  z*inv + (beta - mean*inv),  inv = gamma*rsqrt(var+eps)  (one fma per element), so the result carries one
    fp32 rounding of mean*inv (= 3.25*31.6 ~ 103 here: ulp/2 = 3.8e-6)."""
    L = dclib
    C, M = 32, 512
    z = np.full((M, C), 3.25, np.float32)
    part = np.stack([z.astype(np.float64).sum(0), (z.astype(np.float64) ** 2).sum(0)], -1)[None]
    mean, invstd = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    L.dc_bn_stats_finalize(dev(part).data_ptr(), 1, 1, C, float(M), 1e-3, -1.0, mean.data_ptr(), invstd.data_ptr(), None, None, None)
    beta = np.linspace(-1, 1, C).astype(np.float32)
    out = torch.empty((M, C), device='cuda')
    L.dc_bn_relu_drop_fwd(dev(z).data_ptr(), mean.data_ptr(), invstd.data_ptr(), dev(np.ones(C, np.float32)).data_ptr(),
                          dev(beta).data_ptr(), None, 1.0, 0, out.data_ptr(), C, M, C, 0.0, None, None)
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), np.maximum(beta, 0)[None].repeat(M, 0), atol=5e-6)


def test_dropout_rng_is_reproducible_and_unbiased(dclib):
    L = dclib
    C, M, keep = 64, 4096, 0.75
    z = np.ones((M, C), np.float32)
    one, zero = dev(np.ones(C, np.float32)), dev(np.zeros(C, np.float32))
    outs = []
    for seed in (123, 123, 124):
        out = torch.empty((M, C), device='cuda')
        # mean 0, invstd 1, gamma 1, beta 0 -> y = z = 1
        L.dc_bn_relu_drop_fwd(dev(z).data_ptr(), zero.data_ptr(), one.data_ptr(), one.data_ptr(), zero.data_ptr(), None,
                              keep, seed, out.data_ptr(), C, M, C, 0.0, None, None)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])
    vals = np.unique(outs[0])
    assert np.allclose(vals, [0, 1 / keep])
    assert abs((outs[0] > 0).mean() - keep) < 0.01


@pytest.mark.parametrize('N,H,W,C', [(2, 16, 16, 32), (1, 12, 20, 8), (1, 64, 64, 64)])
def test_maxpool_bit_exact_with_ties(dclib, N, H, W, C):
    L = dclib
    rs = np.random.RandomState(H * C)
    x = np.maximum(rs.standard_normal((N, H, W, C)), 0).astype(np.float32)      # post-ReLU: many exact ties at 0
    x[0, :4, :4] = 1.5                                                         # an all-equal window => index 0
    p_ref, i_ref = on.maxpool2x2_fwd(x)
    cat = np.zeros((N, H, W, 2 * C), np.float32)
    cat[..., C:] = x
    out = torch.empty((N, H // 2, W // 2, C), device='cuda')
    idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device='cuda')
    L.dc_maxpool2x2_fwd(dev(cat).data_ptr() + 4 * C, 2 * C, out.data_ptr(), idx.data_ptr(), N, H, W, C, None)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), p_ref)
    assert np.array_equal(idx.cpu().numpy(), i_ref)                             # bit-exact argmax contract
    assert np.all(idx.cpu().numpy()[0, :2, :2] == 0)
    dy = rs.standard_normal(p_ref.shape).astype(np.float32)
    skip = rs.standard_normal((N, H, W, 2 * C)).astype(np.float32)
    dx = torch.empty((N, H, W, C), device='cuda')
    L.dc_maxpool2x2_bwd(dev(dy).data_ptr(), idx.data_ptr(), dev(skip).data_ptr() + 4 * C, 2 * C, dx.data_ptr(), N, H, W, C, None)
    torch.cuda.synchronize()
    assert np.array_equal(dx.cpu().numpy(), (on.maxpool2x2_bwd(dy, i_ref) + skip[..., C:]).astype(np.float32))
    L.dc_maxpool2x2_bwd(dev(dy).data_ptr(), idx.data_ptr(), None, 0, dx.data_ptr(), N, H, W, C, None)
    torch.cuda.synchronize()
    assert np.array_equal(dx.cpu().numpy(), on.maxpool2x2_bwd(dy, i_ref))


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 64, 64, 64, 64), (1, 40, 72, 48, 80), (2, 48, 48, 32, 32), (1, 32, 32, 128, 256), (3, 33, 50, 64, 96)])
def test_dgrad_with_fused_bn_backward_sums(dclib, N, H, W, Cin, Cout):
    """dc_conv3x3_dgrad_bnred_f16x3: the data gradient is the plain kernel's bit for bit, and the BatchNorm-backward sums
    it emits for the layer in front (gate from that layer's own affine, xhat from its statistics) finalize to what
    dc_bn_bwd_reduce gives on the same da / z -- interior and ragged tiles, 2- / 3- / many-step tiles, both tile shapes."""
    L = dclib
    rows = L.dc_conv3x3_dgrad_bnred_blocks(N, H, W, Cin, Cout)
    assert rows > 0
    rs = np.random.RandomState(Cin + H)
    dz = dev(rs.standard_normal((N, H, W, Cout)).astype(np.float32))
    K = dev((rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32))
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    z = dev(rs.standard_normal((N, H, W, Cin)).astype(np.float32))
    mu = dev((rs.standard_normal(Cin) * 0.2).astype(np.float32)); isd = dev((rs.random_sample(Cin) + 0.5).astype(np.float32))
    ga = dev((rs.standard_normal(Cin)).astype(np.float32)); be = dev((rs.standard_normal(Cin) * 0.3).astype(np.float32))
    scale = torch.full((4,), 4.0, device='cuda')
    dx0 = torch.full((N, H, W, Cin), float('nan'), device='cuda'); dx1 = torch.full_like(dx0, float('nan'))
    L.dc_conv3x3_dgrad_f16x3(dz.data_ptr(), wpd.data_ptr(), dx0.data_ptr(), scale.data_ptr(), None, 0, None, N, H, W, Cin, Cout, None)
    part = torch.full((rows * Cin * 2,), float('nan'), device='cuda')
    amx = torch.full((rows * Cin,), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_bnred_f16x3(dz.data_ptr(), wpd.data_ptr(), dx1.data_ptr(), scale.data_ptr(), None, 0, z.data_ptr(), mu.data_ptr(),
                                   isd.data_ptr(), ga.data_ptr(), be.data_ptr(), part.data_ptr(), amx.data_ptr(), N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    assert np.array_equal(dx0.cpu().numpy(), dx1.cpu().numpy())
    assert np.isfinite(part.cpu().numpy()).all()
    # the same launch with the power of two derived in the kernel from per-block maxima (max 255.9 -> scale 4): same bits
    amax_part = torch.full((19,), 100.0, device='cuda')
    amax_part[7] = 255.9
    dx2, part2 = torch.full_like(dx0, float('nan')), torch.full_like(part, float('nan'))
    L.dc_conv3x3_dgrad_bnred_f16x3(dz.data_ptr(), wpd.data_ptr(), dx2.data_ptr(), None, amax_part.data_ptr(), 19, z.data_ptr(), mu.data_ptr(),
                                   isd.data_ptr(), ga.data_ptr(), be.data_ptr(), part2.data_ptr(), None, N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2) and torch.equal(part, part2)
    pixels = N * H * W
    todo = [(part, rows)]
    if Cin & (Cin - 1) == 0:             # (the two-pass kernel takes power-of-two channel counts only)
        blocks = L.dc_bn_bwd_blocks(pixels, Cin)
        p2 = torch.zeros(blocks * Cin * 2, device='cuda')
        am2 = torch.zeros(blocks * Cin, device='cuda')
        L.dc_bn_bwd_reduce(dx0.data_ptr(), Cin, z.data_ptr(), mu.data_ptr(), isd.data_ptr(), ga.data_ptr(), be.data_ptr(), None, 1.0, 0,
                           p2.data_ptr(), am2.data_ptr(), pixels, Cin, None)
        todo.append((p2, blocks))
    out = []
    for ptr, r in todo:
        dg, db = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
        L.dc_bn_bwd_finalize(ptr.data_ptr(), r, Cin, dg.data_ptr(), db.data_ptr(), None)
        torch.cuda.synchronize()
        out.append((dg.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64)))
    # float64 reference of the sums
    dxa, za = dx0.cpu().numpy().astype(np.float64).reshape(-1, Cin), z.cpu().numpy().astype(np.float64).reshape(-1, Cin)
    sc = (ga.cpu().numpy() * isd.cpu().numpy()).astype(np.float32)               # dc_bn_affine: fp32 product, then ONE fmaf
    sh = (be.cpu().numpy().astype(np.float64) - mu.cpu().numpy().astype(np.float64) * sc.astype(np.float64)).astype(np.float32)
    gate = (za * sc.astype(np.float64) + sh.astype(np.float64)) > 0                 # the sign of the exact fmaf argument
    dy = np.where(gate, dxa, 0.0)
    ref_db = dy.sum(0)
    ref_dg = (dy * (za - mu.cpu().numpy().astype(np.float64)) * isd.cpu().numpy().astype(np.float64)).sum(0)
    tol = 2e-5 * max(np.abs(ref_dg).max(), np.abs(ref_db).max())
    for dg, db in out:
        assert np.abs(dg - ref_dg).max() < tol and np.abs(db - ref_db).max() < tol
    assert L.dc_conv3x3_dgrad_bnred_blocks(2, 16, 16, 64, 64) == 0          # narrow layers stay on the two-pass path


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 64, 64, 64, 64), (1, 40, 72, 48, 80), (2, 48, 64, 32, 32), (8, 128, 128, 64, 128), (3, 34, 50, 64, 96)])
def test_inference_conv_with_pooled_output_equals_conv_then_pool(dclib, N, H, W, Cin, Cout):
    """dc_conv3x3_fwd_pool_f16x3 (folded BN + ReLU, optimistic range flag) == dc_conv3x3_fwd_f16x3 followed by
    dc_maxpool2x2_fwd on its (strided) output, bit for bit -- interior and ragged tiles, both tile shapes."""
    L = dclib
    if not _fused_paths_enabled():
        pytest.skip('pooled-output convolution switched off by the environment')
    assert L.dc_conv3x3_fwd_pool_blocks(N, H, W, Cin, Cout) > 0
    rs = np.random.RandomState(H + Cout)
    x = dev(rs.standard_normal((N, H, W, Cin)).astype(np.float32))
    K = dev((rs.standard_normal((3, 3, Cin, Cout)) * 0.06).astype(np.float32))
    wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cin, Cout), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wp.data_ptr(), 9, Cin, Cout, Cin * Cout, Cout, 1, 0, None)
    sc = dev((rs.random_sample(Cout) + 0.5).astype(np.float32)); sh = dev((rs.standard_normal(Cout) * 0.4).astype(np.float32))
    ld = 2 * Cout
    outs = []
    for fused in (False, True):
        cat = torch.zeros((N, H, W, ld), device='cuda')
        pool = torch.full((N, H // 2, W // 2, Cout), float('nan'), device='cuda')
        flag = torch.zeros(4, device='cuda')
        if fused:
            L.dc_conv3x3_fwd_pool_f16x3(x.data_ptr(), wp.data_ptr(), None, cat.data_ptr() + 4 * Cout, ld, sc.data_ptr(), sh.data_ptr(), 1,
                                        flag.data_ptr(), pool.data_ptr(), N, H, W, Cin, Cout, None)
        else:
            L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp.data_ptr(), None, cat.data_ptr() + 4 * Cout, ld, None, 0, sc.data_ptr(), sh.data_ptr(), 1,
                                   None, 0, flag.data_ptr(), -1, None, N, H, W, Cin, Cout, None)
            L.dc_maxpool2x2_fwd(cat.data_ptr() + 4 * Cout, ld, pool.data_ptr(), None, N, H, W, Cout, None)
        torch.cuda.synchronize()
        outs.append((cat.cpu().numpy(), pool.cpu().numpy(), flag.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.isfinite(outs[1][1]).all() and (outs[1][1] > 0).any() and (outs[1][0][..., :Cout] == 0).all()
    assert L.dc_conv3x3_fwd_pool_blocks(2, 16, 16, 64, 64) == 0 and L.dc_conv3x3_fwd_pool_blocks(2, 63, 64, 64, 64) == 0


@pytest.mark.parametrize('C,pixels,kind,bnin', [(32, 5000, 0, True), (8, 777, 1, True), (4, 64, 0, False), (64, 4097, 0, True), (16, 300, 1, False)])
def test_head_fused_forward_backward_equals_separate_calls(dclib, C, pixels, kind, bnin):
    """dc_head_fwd_bwd (training with a per-pixel loss) against dc_head_fwd followed by dc_head_bwd(_bnin_bnred): p, da
    and the metric partial sums bit for bit (same per-pixel arithmetic), the head's weight gradient and the producing
    BatchNorm layer's backward sums to summation-order accuracy (same terms, different order inside a block)."""
    L = dclib
    rs = np.random.RandomState(C + pixels)
    a = dev((rs.standard_normal((pixels, C)) * 1.5).astype(np.float32))
    kh = dev((rs.standard_normal((C, 2)) * 0.4).astype(np.float32)); bh = dev(rs.standard_normal(2).astype(np.float32) * 0.1)
    y = dev((rs.random_sample(pixels) < 0.2).astype(np.uint8))
    sc = dev((rs.random_sample(C) + 0.5).astype(np.float32)); sh = dev((rs.standard_normal(C) * 0.3).astype(np.float32))
    mu = dev(rs.standard_normal(C).astype(np.float32) * 0.2); isd = dev((rs.random_sample(C) + 0.5).astype(np.float32))
    hb = L.dc_head_blocks(pixels)
    def bufs():
        return dict(p=torch.empty(pixels, device='cuda'), part=torch.zeros(hb * 12, device='cuda'), da=torch.empty((pixels, C), device='cuda'),
                    gp=torch.zeros(hb * (C + 4), device='cuda'), bp=torch.zeros(hb * C * 2, device='cuda'),
                    sums=torch.zeros(12, dtype=torch.float64, device='cuda'), dk=torch.zeros((C, 2), device='cuda'), db=torch.zeros(2, device='cuda'),
                    dg=torch.zeros(C, device='cuda'), dbt=torch.zeros(C, device='cuda'), am=torch.zeros(hb * C, device='cuda'))
    A, B = bufs(), bufs()
    scp, shp = (sc.data_ptr(), sh.data_ptr()) if bnin else (None, None)
    # separate
    if bnin:
        L.dc_head_fwd_bnin(a.data_ptr(), scp, shp, kh.data_ptr(), bh.data_ptr(), y.data_ptr(), A['p'].data_ptr(), A['part'].data_ptr(), pixels, C, None)
        L.dc_reduce_partials_f64(A['part'].data_ptr(), hb, 12, A['sums'].data_ptr(), None)
        L.dc_head_bwd_bnin_bnred(a.data_ptr(), scp, shp, A['p'].data_ptr(), y.data_ptr(), kh.data_ptr(), A['da'].data_ptr(), A['gp'].data_ptr(), kind,
                                 A['sums'].data_ptr(), mu.data_ptr(), isd.data_ptr(), A['bp'].data_ptr(), A['am'].data_ptr(), pixels, C, None)
    else:
        L.dc_head_fwd(a.data_ptr(), kh.data_ptr(), bh.data_ptr(), y.data_ptr(), A['p'].data_ptr(), A['part'].data_ptr(), pixels, C, None)
        L.dc_reduce_partials_f64(A['part'].data_ptr(), hb, 12, A['sums'].data_ptr(), None)
        L.dc_head_bwd(a.data_ptr(), A['p'].data_ptr(), y.data_ptr(), kh.data_ptr(), A['da'].data_ptr(), A['gp'].data_ptr(), kind, A['sums'].data_ptr(), pixels, C, None)
    # fused
    L.dc_head_fwd_bwd(a.data_ptr(), scp, shp, kh.data_ptr(), bh.data_ptr(), y.data_ptr(), B['p'].data_ptr(), B['part'].data_ptr(), B['da'].data_ptr(),
                      B['gp'].data_ptr(), kind, mu.data_ptr() if bnin else None, isd.data_ptr() if bnin else None, B['bp'].data_ptr() if bnin else None,
                      B['am'].data_ptr() if bnin else None, pixels, C, None)
    L.dc_reduce_partials_f64(B['part'].data_ptr(), hb, 12, B['sums'].data_ptr(), None)
    for X in (A, B):
        L.dc_head_grad_finalize(X['gp'].data_ptr(), hb, C, X['dk'].data_ptr(), X['db'].data_ptr(), None)
        if bnin:
            L.dc_bn_bwd_finalize(X['bp'].data_ptr(), hb, C, X['dg'].data_ptr(), X['dbt'].data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(A['p'].cpu().numpy(), B['p'].cpu().numpy())
    assert np.array_equal(A['da'].cpu().numpy(), B['da'].cpu().numpy())
    assert np.array_equal(A['sums'].cpu().numpy(), B['sums'].cpu().numpy())
    if bnin:       # per-(block, channel) max |da| (the dz-on-load bound's input): its maximum over the blocks is the tensor's
        amax_ref = np.abs(A['da'].cpu().numpy()).max(0)
        for X in (A, B):
            am = X['am'].cpu().numpy().reshape(hb, C).max(0)
            assert np.all(am >= amax_ref * (1 - 1e-6)) and np.all(am <= amax_ref * (1 + 1e-6) + 1e-30), (am, amax_ref)
    for k in ('dk', 'db') + (('dg', 'dbt') if bnin else ()):
        x, z = A[k].cpu().numpy().astype(np.float64), B[k].cpu().numpy().astype(np.float64)
        assert np.abs(x - z).max() <= 2e-6 * max(np.abs(x).max(), 1e-30), (k, x, z)
    with pytest.raises(Exception):
        L.dc_head_fwd_bwd(a.data_ptr(), scp, shp, kh.data_ptr(), bh.data_ptr(), y.data_ptr(), B['p'].data_ptr(), B['part'].data_ptr(), B['da'].data_ptr(),
                          B['gp'].data_ptr(), 2, None, None, None, None, pixels, C, None)


@pytest.mark.parametrize('N,H,W,C,drop', [(2, 16, 16, 32, 'rng'), (1, 12, 20, 8, 'mask'), (3, 8, 8, 64, 'none'), (1, 64, 64, 256, 'rng')])
def test_bn_relu_drop_pool_fused_equals_separate_calls(dclib, N, H, W, C, drop):
    """dc_bn_relu_drop_pool_fwd == dc_bn_relu_drop_fwd followed by dc_maxpool2x2_fwd, bit for bit: activation (strided
    into a concat buffer), pooled values, argmax (post-ReLU / dropout zeros give plenty of ties), range-guard bound."""
    L = dclib
    rs = np.random.RandomState(C + H)
    z = dev(rs.standard_normal((N, H, W, C)).astype(np.float32) * 2)
    v = [dev(a.astype(np.float32)) for a in (rs.standard_normal(C), rs.random_sample(C) + 0.5, rs.standard_normal(C), rs.standard_normal(C) * 0.3)]
    mask = dev((rs.random_sample((N, H, W, C)) < 0.75).astype(np.uint8)) if drop == 'mask' else None
    keep = 1.0 if drop == 'none' else 0.75
    seed = 0x1234567 if drop == 'rng' else 0
    mptr = mask.data_ptr() if mask is not None else None
    ld = 2 * C
    res = []
    for fused in (False, True):
        cat = torch.zeros((N, H, W, ld), device='cuda')
        pooled = torch.empty((N, H // 2, W // 2, C), device='cuda')
        idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device='cuda')
        ab = torch.zeros(C, device='cuda')
        if fused:
            L.dc_bn_relu_drop_pool_fwd(z.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), mptr, keep, seed,
                                       cat.data_ptr() + 4 * C, ld, pooled.data_ptr(), idx.data_ptr(), N, H, W, C, float(N * H * W),
                                       ab.data_ptr(), None)
        else:
            L.dc_bn_relu_drop_fwd(z.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), mptr, keep, seed,
                                  cat.data_ptr() + 4 * C, ld, N * H * W, C, float(N * H * W), ab.data_ptr(), None)
            L.dc_maxpool2x2_fwd(cat.data_ptr() + 4 * C, ld, pooled.data_ptr(), idx.data_ptr(), N, H, W, C, None)
        torch.cuda.synchronize()
        res.append([t.cpu().numpy() for t in (cat, pooled, idx, ab)])
    for a, b, what in zip(res[0], res[1], ('activation', 'pooled', 'argmax', 'bound')):
        assert np.array_equal(a, b), what
    assert (res[1][0][..., :C] == 0).all() and (res[1][2] > 0).any() and (res[1][1] == 0).any()


@pytest.mark.parametrize('C,pixels', [(32, 5000), (8, 777), (4, 64)])
def test_head_fwd_bwd_metrics(dclib, C, pixels):
    L = dclib
    rs = np.random.RandomState(C + pixels)
    a = np.maximum(rs.standard_normal((pixels, C)), 0).astype(np.float32)
    a[:5] = 0                                           # logits == bias == 0 -> p exactly 0.5 (round-half-even, BCE kink)
    a[5:8] *= 60                                        # saturate: p hits the 1e-7 clip
    Kh = rs.standard_normal((1, 1, C, 2)).astype(np.float32)
    bh = np.zeros(2, np.float32)
    y = (rs.random_sample(pixels) < 0.2).astype(np.uint8)
    p_ref, sm = on.head_fwd(a.astype(np.float64), Kh.astype(np.float64), bh.astype(np.float64))
    blocks = L.dc_head_blocks(pixels)
    p = torch.empty(pixels, device='cuda')
    part = torch.empty(blocks * 12, device='cuda')
    sums = torch.empty(12, dtype=torch.float64, device='cuda')
    ad, yd, kd = dev(a), dev(y), dev(Kh)
    L.dc_head_fwd(ad.data_ptr(), kd.data_ptr(), dev(bh).data_ptr(), yd.data_ptr(), p.data_ptr(), part.data_ptr(), pixels, C, None)
    L.dc_reduce_partials_f64(part.data_ptr(), blocks, 12, sums.data_ptr(), None)
    torch.cuda.synchronize()
    pg = p.cpu().numpy()
    assert np.abs(pg - p_ref).max() < 1e-6
    assert np.all(pg[:5] == 0.5)
    s = sums.cpu().numpy()
    yf = y.astype(np.float64)
    assert abs(s[0] / pixels - on.bce_keras(pg.astype(np.float64), yf)) < 1e-6     # loss on the GPU's own p
    # vs the float64 p: fp32 rounding of p near 1 moves -log(1-p) by a few % on near-saturated pixels
    assert abs(s[0] / pixels - on.bce_keras(p_ref, yf)) < 1e-4
    pr = np.round(pg.astype(np.float64))
    exp = [None, (pr * yf).sum(), pr.sum(), np.clip(yf - pr, 0, 1).sum(), yf.sum(), (yf * pg).sum(), (pg.astype(np.float64) ** 2).sum(), (yf ** 2).sum()]
    for k in range(1, 8):
        assert abs(s[k] - exp[k]) < 1e-3 * max(1.0, abs(exp[k])), k
    # backward
    da = torch.empty((pixels, C), device='cuda')
    part2 = torch.empty(blocks * (C + 4), device='cuda')
    dk, db = torch.empty((1, 1, C, 2), device='cuda'), torch.empty(2, device='cuda')
    L.dc_head_bwd(ad.data_ptr(), p.data_ptr(), yd.data_ptr(), kd.data_ptr(), da.data_ptr(), part2.data_ptr(), 0, sums.data_ptr(), pixels, C, None)
    L.dc_head_grad_finalize(part2.data_ptr(), blocks, C, dk.data_ptr(), db.data_ptr(), None)
    torch.cuda.synchronize()
    # the backward consumes the stored fp32 p: evaluate the oracle's formula at that p (a saturated pixel whose
    # fp32 p lands exactly on the clip bound is 'clipped' in fp32 and 'inside' in float64)
    p64 = pg.astype(np.float64)
    dp = on.bce_keras_grad(p64, yf)
    sref = dp * p64 * (1 - p64)
    dlog = np.stack([-sref, sref], -1)
    assert np.abs(da.cpu().numpy() - dlog @ Kh[0, 0].astype(np.float64).T).max() < 1e-6
    assert np.abs(dk.cpu().numpy()[0, 0] - a.astype(np.float64).T @ dlog).max() < 2e-5
    assert np.abs(db.cpu().numpy() - dlog.sum(0)).max() < 2e-5
    # alternate losses (unet_2d_summary.py:372-377): value from the sums, gradient through the same kernel
    from deep_calcium_amd.model import metrics_from_sums, LOSS_KINDS
    assert abs(s[8] - p64.sum()) < 1e-3 * p64.sum()
    for name, kind in LOSS_KINDS.items():
        if kind == 0:
            continue
        l_ref, dp_ref = on.alt_loss(name, p64, yf)
        assert abs(metrics_from_sums(s, pixels, name)['loss'] - l_ref) < 2e-5 * max(1.0, abs(l_ref)), name
        L.dc_head_bwd(ad.data_ptr(), p.data_ptr(), yd.data_ptr(), kd.data_ptr(), da.data_ptr(), part2.data_ptr(), kind,
                      sums.data_ptr(), pixels, C, None)
        torch.cuda.synchronize()
        sref = dp_ref * p64 * (1 - p64)
        ref = np.stack([-sref, sref], -1) @ Kh[0, 0].astype(np.float64).T
        assert np.abs(da.cpu().numpy() - ref).max() < 2e-5 * max(np.abs(ref).max(), 1e-12), name


@pytest.mark.parametrize('N,H,W,C,keep', [(2, 16, 16, 32, 1.0), (1, 32, 48, 64, 0.75), (3, 8, 8, 8, 0.5)])
def test_bnred_sums_from_pool_and_head_backward(dclib, N, H, W, C, keep):
    """"bnred" entry points: the kernel that writes da also emits the BatchNorm-backward pass-1 sums; after
    dc_bn_bwd_finalize they must equal what dc_bn_bwd_reduce computes from the da the same kernel wrote."""
    L = dclib
    rs = np.random.RandomState(N + H + C)
    z = rs.standard_normal((N, H, W, C)).astype(np.float32)
    gamma = (rs.random_sample(C) + 0.5).astype(np.float32); beta = (rs.standard_normal(C) * 0.3).astype(np.float32)
    mean = z.reshape(-1, C).mean(0).astype(np.float32)
    invstd = (1 / np.sqrt(z.reshape(-1, C).var(0) + 1e-3)).astype(np.float32)
    mask = (rs.random_sample(z.shape) < keep).astype(np.uint8) if keep < 1 else None
    zd, md, isd, gd, bd = dev(z), dev(mean), dev(invstd), dev(gamma), dev(beta)
    mk = dev(mask).data_ptr() if mask is not None else None

    def finalize(partial, P):
        dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        L.dc_bn_bwd_finalize(partial.data_ptr(), P, C, dg.data_ptr(), db.data_ptr(), None)
        torch.cuda.synchronize()
        return dg.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64)

    # ---- max-pool backward
    dy = rs.standard_normal((N, H // 2, W // 2, C)).astype(np.float32)
    idx = rs.randint(0, 4, dy.shape).astype(np.uint8)
    skip = rs.standard_normal((N, H, W, C + 8)).astype(np.float32)
    dyd, idd, skd = dev(dy), dev(idx), dev(skip)
    dx1, dx2 = torch.empty(N, H, W, C, device='cuda'), torch.empty(N, H, W, C, device='cuda')
    L.dc_maxpool2x2_bwd(dyd.data_ptr(), idd.data_ptr(), skd.data_ptr() + 32, C + 8, dx1.data_ptr(), N, H, W, C, None)
    P = L.dc_maxpool2x2_bwd_blocks(N, H, W, C)
    part = torch.full((P * C * 2,), float('nan'), device='cuda')
    amx = torch.full((P * C,), float('nan'), device='cuda')
    L.dc_maxpool2x2_bwd_bnred(dyd.data_ptr(), idd.data_ptr(), skd.data_ptr() + 32, C + 8, dx2.data_ptr(), zd.data_ptr(),
                              md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), mk, keep, 0,
                              part.data_ptr(), amx.data_ptr(), N, H, W, C, None)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2)
    Pr = L.dc_bn_bwd_blocks(N * H * W, C)
    part_r = torch.empty(Pr * C * 2, device='cuda')
    amx_r = torch.full((Pr * C,), float('nan'), device='cuda')
    L.dc_bn_bwd_reduce(dx1.data_ptr(), C, zd.data_ptr(), md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), mk,
                       keep, 0, part_r.data_ptr(), amx_r.data_ptr(), N * H * W, C, None)
    (dg, db), (dg_r, db_r) = finalize(part, P), finalize(part_r, Pr)
    tol = 1e-5 * max(1.0, np.abs(dg_r).max(), np.abs(db_r).max())
    assert np.abs(dg - dg_r).max() < tol and np.abs(db - db_r).max() < tol
    # per-(row, channel) max |dy| (dy = da * relu gate * dropout factor): both producers' maxima over their rows are the
    # tensor's per-channel maximum, exactly (a max has no rounding)
    sc32 = gamma * invstd
    sh32 = (beta.astype(np.float64) - mean.astype(np.float64) * sc32.astype(np.float64)).astype(np.float32)
    gate = (z.astype(np.float64) * sc32.astype(np.float64) + sh32.astype(np.float64)) > 0
    dy_ref = dx1.cpu().numpy() * gate * ((mask.astype(np.float32) * np.float32(1.0 / keep)) if mask is not None else 1.0)
    amax_ref = np.abs(dy_ref).reshape(-1, C).max(0)
    for a_, rows_ in ((amx, P), (amx_r, Pr)):
        got = a_.cpu().numpy().reshape(rows_, C).max(0)
        assert np.allclose(got, amax_ref, rtol=1e-6, atol=0), (got, amax_ref)

    # ---- head backward (BN + ReLU on load, no dropout on the head's input layer)
    if keep == 1.0:
        pixels = N * H * W
        sc, sh = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        part0 = np.stack([z.reshape(-1, C).astype(np.float64).sum(0), (z.reshape(-1, C).astype(np.float64) ** 2).sum(0)], -1)
        L.dc_bn_stats_finalize_affine(dev(part0).data_ptr(), 1, 1, C, float(pixels), 1e-3, -1.0,
                                      md.data_ptr(), isd.data_ptr(), None, None, gd.data_ptr(), bd.data_ptr(),
                                      sc.data_ptr(), sh.data_ptr(), None, None)
        kh = (rs.standard_normal((C, 2)) * 0.3).astype(np.float32)
        y = (rs.random_sample(pixels) < 0.3).astype(np.uint8)
        p = rs.random_sample(pixels).astype(np.float32)
        hb = L.dc_head_blocks(pixels)
        da1, da2 = torch.empty(pixels, C, device='cuda'), torch.empty(pixels, C, device='cuda')
        r1, r2 = torch.zeros(hb * (C + 4), device='cuda'), torch.zeros(hb * (C + 4), device='cuda')
        partb = torch.full((hb * C * 2,), float('nan'), device='cuda')
        khd, yd, pd = dev(kh), dev(y), dev(p)
        L.dc_head_bwd_bnin(zd.data_ptr(), sc.data_ptr(), sh.data_ptr(), pd.data_ptr(), yd.data_ptr(), khd.data_ptr(),
                           da1.data_ptr(), r1.data_ptr(), 0, None, pixels, C, None)
        L.dc_head_bwd_bnin_bnred(zd.data_ptr(), sc.data_ptr(), sh.data_ptr(), pd.data_ptr(), yd.data_ptr(), khd.data_ptr(),
                                 da2.data_ptr(), r2.data_ptr(), 0, None, md.data_ptr(), isd.data_ptr(), partb.data_ptr(),
                                 None, pixels, C, None)
        torch.cuda.synchronize()
        assert torch.equal(da1, da2) and torch.equal(r1, r2)
        L.dc_bn_bwd_reduce(da1.data_ptr(), C, zd.data_ptr(), md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                           None, 1.0, 0, part_r.data_ptr(), None, pixels, C, None)
        (dg, db), (dg_r, db_r) = finalize(partb, hb), finalize(part_r, Pr)
        tol = 1e-5 * max(1e-6, np.abs(dg_r).max(), np.abs(db_r).max())
        assert np.abs(dg - dg_r).max() < tol and np.abs(db - db_r).max() < tol


def test_adam_keras_form(dclib):
    L = dclib
    rs = np.random.RandomState(0)
    n = 10007
    p, g = rs.standard_normal(n).astype(np.float32), (rs.standard_normal(n) * 1e-3).astype(np.float32)
    g[:3] = 1.0
    pad = (n + 3) // 4 * 4
    bufs = [torch.zeros(pad, device='cuda') for _ in range(4)]
    bufs[0][:n] = dev(p)
    bufs[1][:n] = dev(g)
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for it in range(3):
        lr_t = 0.002 * np.sqrt(1 - 0.999 ** (it + 1)) / (1 - 0.9 ** (it + 1))
        L.dc_adam_step_flat(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), n, float(lr_t),
                            0.9, 0.999, 1e-8, 1.0, None)
        pr, mr, vr = on.adam_keras(pr, g.astype(np.float64), mr, vr, it)
    torch.cuda.synchronize()
    assert np.abs(bufs[0].cpu().numpy()[:n] - pr).max() < 1e-6
    # closed form of step 1 on g = 1: delta = -lr_t * (1-b1) / (sqrt(1-b2) + 1e-8), lr_t = lr*sqrt(1-b2)/(1-b1)
    d1 = -0.002 * np.sqrt(1 - 0.999) / (1 - 0.9) * 0.1 / (np.sqrt(0.001) + 1e-8)
    assert abs(d1 + 0.002) < 1e-8


def test_reduce_partials_deterministic(dclib):
    L = dclib
    rs = np.random.RandomState(1)
    for P, Ln in ((1, 100), (63, 9216), (64, 300), (2048, 1000)):
        x = rs.standard_normal((P, Ln)).astype(np.float32)
        out1, out2 = torch.empty(Ln, device='cuda'), torch.empty(Ln, device='cuda')
        tmp = torch.empty(32 * Ln, device='cuda')
        xd = dev(x)
        L.dc_reduce_partials(xd.data_ptr(), P, Ln, 0.5, out1.data_ptr(), tmp.data_ptr(), None)
        L.dc_reduce_partials(xd.data_ptr(), P, Ln, 0.5, out2.data_ptr(), tmp.data_ptr(), None)
        torch.cuda.synchronize()
        assert np.array_equal(out1.cpu().numpy(), out2.cpu().numpy())
        assert np.allclose(out1.cpu().numpy(), 0.5 * x.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('N,H,W,Ci,Co,bias_sigmas', [(2, 64, 64, 64, 64, 0), (3, 40, 72, 128, 96, 0), (16, 128, 128, 64, 64, 1000),
                                                     (1, 32, 32, 64, 32, 0), (16, 256, 256, 64, 64, 0), (4, 64, 96, 256, 512, 3)])
def test_conv3x3_bn_partials_per_workgroup(dclib, N, H, W, Ci, Co, bias_sigmas):
    """dc_conv3x3_stats_rows(): where the persistent role-split kernel serves the forward launch every (workgroup, consumer set)
    Chan-merges its tiles and writes ONE row of BatchNorm partials.  Same z bit for bit, at most 2 x #CUs rows, the statistics
    they finalize to equal the per-tile rows' (and the float64 oracle's) -- also at |mean| = 1000 sigma --, bit-reproducible."""
    L = dclib
    rs = np.random.RandomState(N + H + Co)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    K = (rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    b = (rs.standard_normal(Co) + bias_sigmas * np.sqrt(2.0)).astype(np.float32)
    xd, bd = dev(x), dev(b)
    wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
    L.dc_pack_weights_f16x3(dev(K).data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
    tiles, rows = L.dc_conv3x3_tiles(N, H, W, Co), L.dc_conv3x3_stats_rows(N, H, W, Ci, Co)
    assert 0 < rows <= tiles
    if L.dc_conv3x3_pp_blocks(N, H, W, Ci, Co, 0, 1) > 0:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert rows == 2 * min(cus, L.dc_conv3x3_pp_blocks(N, H, W, Ci, Co, 0, 1) * ((Co + 63) // 64 if Co > 32 else 1)) or rows == tiles
    else:
        assert rows == tiles

    def run(stats_rows):
        z = torch.full((N, H, W, Co), float('nan'), device='cuda')
        st = torch.full((max(stats_rows, tiles) * Co * 2,), float('nan'), device='cuda', dtype=torch.float64)
        L.dc_conv3x3_fwd_f16x3(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), z.data_ptr(), Co, st.data_ptr(), stats_rows, None, None, 0,
                               None, 0, None, 0, None, N, H, W, Ci, Co, None)
        n = stats_rows if stats_rows else tiles
        mean, invstd = torch.empty(Co, device='cuda'), torch.empty(Co, device='cuda')
        mm, mv = torch.zeros(Co, device='cuda'), torch.ones(Co, device='cuda')
        L.dc_bn_stats_finalize(st.data_ptr(), n, 1, Co, float(N * H * W), 1e-3, 0.99, mean.data_ptr(), invstd.data_ptr(), mm.data_ptr(),
                               mv.data_ptr(), None)
        torch.cuda.synchronize()
        assert torch.isfinite(st[:n * Co * 2]).all()
        return z, st[:n * Co * 2].clone(), mean.cpu().numpy().astype(np.float64), invstd.cpu().numpy().astype(np.float64)

    z0, _, mean0, is0 = run(0)
    z1, st1, mean1, is1 = run(rows)
    z2, st2, _, _ = run(rows)
    assert torch.equal(z0, z1) and torch.equal(st1, st2) and torch.equal(z1, z2)
    zr = z0.cpu().numpy().astype(np.float64).reshape(-1, Co)
    mu, var = zr.mean(0), zr.var(0)
    isr = 1.0 / np.sqrt(var + 1e-3)
    tol_m = 2e-6 * np.abs(mu).max() + 1e-6 * np.sqrt(var).max()
    assert np.abs(mean1 - mu).max() < tol_m and np.abs(mean0 - mu).max() < tol_m
    assert np.abs(is1 / isr - 1).max() < 2e-5 and np.abs(is0 / isr - 1).max() < 2e-5
    with pytest.raises(Exception, match='stats_rows'):
        L.dc_conv3x3_fwd_f16x3(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), z0.data_ptr(), Co, st1.data_ptr(), 7, None, None, 0,
                               None, 0, None, 0, None, N, H, W, Ci, Co, None)


def test_split_k_workspace_is_caller_owned_one_per_stream(dclib):
    """Split-K launches (narrow layers) keep their slabs in the CALLER's workspace (dc_conv3x3_splitk_ws_floats): the same
    convolution issued on two streams at once, each with its own workspace, then a larger one, must give the bits of the
    same launches issued alone, one after the other -- and the library itself never allocates (tests/test_abi.py)."""
    L = dclib
    rs = np.random.RandomState(11)
    shapes = [(20, 16, 16, 256, 256), (24, 16, 16, 512, 256)]
    data = []
    for N, H, W, Ci, Co in shapes:
        x = dev(rs.standard_normal((N, H, W, Ci)).astype(np.float32))
        K = dev((rs.standard_normal((3, 3, Ci, Co)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32))
        b = dev(rs.standard_normal(Co).astype(np.float32))
        wp = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
        L.dc_pack_weights_f16x3(K.data_ptr(), wp.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
        data.append((x, wp, b))
    need = max(L.dc_conv3x3_splitk_ws_floats(*sh, 0) for sh in shapes)
    assert need > 0
    wss = [torch.full((need,), float('nan'), device='cuda') for _ in range(3)]
    torch.cuda.synchronize()

    def run(i, stream, ws):
        N, H, W, Ci, Co = shapes[i]
        x, wp, b = data[i]
        tiles = L.dc_conv3x3_tiles(N, H, W, Co)
        z = torch.full((N, H, W, Co), float('nan'), device='cuda')
        stats = torch.full((tiles * Co * 2,), float('nan'), device='cuda', dtype=torch.float64)
        if stream is not None:          # the NaN fills run on the current stream; torch's pool streams do not synchronise with it implicitly
            stream.wait_stream(torch.cuda.current_stream())
        L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp.data_ptr(), b.data_ptr(), z.data_ptr(), Co, stats.data_ptr(), 0, None, None, 0,
                               None, 0, None, 0, ws.data_ptr(), N, H, W, Ci, Co, stream.cuda_stream if stream is not None else None)
        return z, stats

    ref = [run(0, None, wss[0]), run(1, None, wss[0])]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        outs.append((run(0, s1, wss[1]), run(0, s2, wss[2])))      # both streams busy with the same shape, one workspace each
    big = run(1, s1, wss[1])
    again = run(0, s1, wss[1])
    torch.cuda.synchronize()
    for a, b2 in outs:
        for z, st in (a, b2):
            assert torch.equal(z, ref[0][0]) and torch.equal(st, ref[0][1])
    assert torch.equal(big[0], ref[1][0]) and torch.equal(big[1], ref[1][1])
    assert torch.equal(again[0], ref[0][0]) and torch.equal(again[1], ref[0][1])
    from deep_calcium_amd._lib import DcunetError
    with pytest.raises(DcunetError, match='16-byte aligned'):
        N, H, W, Ci, Co = shapes[0]
        x, wp, b = data[0]
        z = torch.empty((N, H, W, Co), device='cuda')
        L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp.data_ptr(), b.data_ptr(), z.data_ptr(), Co, None, 0, None, None, 0,
                               None, 0, None, 0, wss[0].data_ptr() + 4, N, H, W, Ci, Co, None)


def test_error_reporting(dclib):
    from deep_calcium_amd._lib import DcunetError
    with pytest.raises(DcunetError, match='null pointer'):
        dclib.dc_conv3x3_dgrad(None, None, None, 1, 8, 8, 8, 8, None)
    x = torch.empty(64, device='cuda')
    with pytest.raises(DcunetError, match='multiple of 4'):
        dclib.dc_conv3x3_dgrad(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 8, 8, 8, 6, None)


@pytest.mark.parametrize('N,H,W,Ci,Co', [(2, 32, 32, 64, 32), (1, 20, 40, 128, 64), (2, 17, 33, 256, 128), (1, 64, 64, 64, 32)])
def test_convT_dgrad_with_fused_bn_backward_sums(dclib, N, H, W, Ci, Co):
    """dc_convT2x2_dgrad_bnred_f16x3: the data gradient is dc_convT2x2_dgrad_f16x3's bit for bit, and the pass-1 sums / max |dy| it
    emits for the block in front of the up-convolution finalize to a float64 reduction of the dx it wrote (both column-block
    widths, ragged tiles)."""
    L = dclib
    rows = L.dc_convT2x2_dgrad_bnred_blocks(N, H, W, Ci, Co)
    assert rows == N * ((W + 31) // 32) * ((H + 3) // 4)
    assert L.dc_convT2x2_dgrad_bnred_blocks(N, 16, 16, Ci, Co) == 0
    rs = np.random.RandomState(Ci + W)
    dz = dev((rs.standard_normal((N, 2 * H, 2 * W, Co)) * 1e-3).astype(np.float32))
    K = dev((rs.standard_normal((2, 2, Co, Ci)) * 0.05).astype(np.float32))
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(4, Co, Ci), device='cuda')
    L.dc_pack_weights_f16x3(K.data_ptr(), wpd.data_ptr(), 4, Co, Ci, Co * Ci, Ci, 1, 0, None)
    z = dev(rs.standard_normal((N, H, W, Ci)).astype(np.float32))
    mu = dev((rs.standard_normal(Ci) * 0.2).astype(np.float32)); isd = dev((rs.random_sample(Ci) + 0.5).astype(np.float32))
    ga = dev(rs.standard_normal(Ci).astype(np.float32)); be = dev((rs.standard_normal(Ci) * 0.3).astype(np.float32))
    scale = torch.full((4,), 2.0 ** 17, device='cuda')
    dx0 = torch.full((N, H, W, Ci), float('nan'), device='cuda'); dx1 = torch.full_like(dx0, float('nan'))
    L.dc_convT2x2_dgrad_f16x3(dz.data_ptr(), wpd.data_ptr(), dx0.data_ptr(), scale.data_ptr(), None, 0, None, N, H, W, Ci, Co, None)
    part = torch.full((rows * Ci * 2,), float('nan'), device='cuda')
    amx = torch.full((rows * Ci,), float('nan'), device='cuda')
    L.dc_convT2x2_dgrad_bnred_f16x3(dz.data_ptr(), wpd.data_ptr(), dx1.data_ptr(), scale.data_ptr(), None, 0, z.data_ptr(), mu.data_ptr(),
                                    isd.data_ptr(), ga.data_ptr(), be.data_ptr(), part.data_ptr(), amx.data_ptr(), N, H, W, Ci, Co, None)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    assert np.isfinite(part.cpu().numpy()).all() and np.isfinite(amx.cpu().numpy()).all()
    dg, db = torch.zeros(Ci, device='cuda'), torch.zeros(Ci, device='cuda')
    L.dc_bn_bwd_finalize(part.data_ptr(), rows, Ci, dg.data_ptr(), db.data_ptr(), None)
    torch.cuda.synchronize()
    dxa, za = dx0.cpu().numpy().astype(np.float64).reshape(-1, Ci), z.cpu().numpy().astype(np.float64).reshape(-1, Ci)
    sc = (ga.cpu().numpy() * isd.cpu().numpy()).astype(np.float32)
    sh = (be.cpu().numpy().astype(np.float64) - mu.cpu().numpy().astype(np.float64) * sc.astype(np.float64)).astype(np.float32)
    gate = (za * sc.astype(np.float64) + sh.astype(np.float64)) > 0
    dy = np.where(gate, dxa, 0.0)
    ref_db = dy.sum(0)
    ref_dg = (dy * (za - mu.cpu().numpy().astype(np.float64)) * isd.cpu().numpy().astype(np.float64)).sum(0)
    tol = 2e-5 * max(np.abs(ref_dg).max(), np.abs(ref_db).max())
    assert np.abs(dg.cpu().numpy() - ref_dg).max() < tol and np.abs(db.cpu().numpy() - ref_db).max() < tol
    assert np.array_equal(amx.cpu().numpy().reshape(rows, Ci).max(0), np.abs(dy).max(0).astype(np.float32))


def _pack_ref(src, taps, K, Ncols, s_tap, s_k, s_n, flip):
    """numpy restatement of the split-fp16 weight image (csrc/igemm_f16x3.hip): [tap][K/8][hi|lo][n][8 halfs] of
    w_scale * src[(flip ? taps-1-tap : tap)*s_tap + k*s_k + n*s_n], w_scale = 2^(10 - floor(log2 max|src|)); trailer {w_scale, max|src|}."""
    flat = np.asarray(src, np.float32).ravel()
    amax = np.float32(np.abs(flat).max())
    ws = np.float32(2.0 ** (10 - int(np.floor(np.log2(amax))))) if amax > 0 else np.float32(1.0)
    K8 = (K + 7) // 8
    img = np.zeros((taps, K8, 2, Ncols, 8), np.float16)
    for tap in range(taps):
        ts = taps - 1 - tap if flip else tap
        for k in range(K):
            x = flat[ts * s_tap + k * s_k + np.arange(Ncols) * s_n] * ws
            hi = x.astype(np.float16)
            img[tap, k // 8, 0, :, k % 8] = hi
            img[tap, k // 8, 1, :, k % 8] = (x - hi.astype(np.float32)).astype(np.float16)
    return img.ravel(), ws, amax


def test_pack_weights_f16x3_image_single_and_batched(dclib):
    """The split-fp16 weight image bit for bit against its numpy restatement -- the single-job entry point and the batched one (whose
    consecutive jobs over ONE source share the max|src| sweep): forward + data-gradient (flipped, transposed) images of a 3x3 layer, K
    not a multiple of 8, a conv-transpose pair, and a lone job behind them."""
    L = dclib
    rs = np.random.RandomState(5)
    jobs, srcs = [], []

    def layer(kh, ci, co, scale):
        w = dev((rs.standard_normal((kh * kh, ci, co)) * scale).astype(np.float32))
        srcs.append(w)
        return w

    w1 = layer(3, 12, 40, 0.02)                    # K = 12: half-filled last slot in the forward image
    jobs.append((w1, 9, 12, 40, 12 * 40, 40, 1, 0))
    jobs.append((w1, 9, 40, 12, 12 * 40, 1, 40, 1))
    w2 = layer(2, 16, 8, 3e-4)                     # conv-transpose (2,2,Cout,Cin) = [tap][co=16][ci=8] here: forward [1][ci][4 co], dgrad [4][co][ci]
    jobs.append((w2, 1, 8, 4 * 16, 0, 1, 8, 0))
    jobs.append((w2, 4, 16, 8, 16 * 8, 8, 1, 0))
    w3 = layer(3, 32, 32, 5.0)
    jobs.append((w3, 9, 32, 32, 32 * 32, 32, 1, 0))
    single, batched, rows, b = [], [], [], 0
    for (w, taps, K, nc, st, sk, sn, flip) in jobs:
        n = L.dc_pack_weights_f16x3_floats(taps, K, nc)
        d1 = torch.zeros(n, dtype=torch.float32, device='cuda')
        d2 = torch.zeros(n, dtype=torch.float32, device='cuda')
        L.dc_pack_weights_f16x3(w.data_ptr(), d1.data_ptr(), taps, K, nc, st, sk, sn, flip, None)
        single.append(d1)
        batched.append(d2)
        total = taps * ((K + 7) // 8) * 8 * nc
        rows.append([w.data_ptr(), d2.data_ptr(), taps, K, nc, st, sk, sn, flip, b])
        b += max(1, min((total // 8 + 1023) // 1024, 2048))
    rows.append([0] * 9 + [b])
    tab = torch.tensor(rows, dtype=torch.int64, device='cuda')
    L.dc_pack_weights_f16x3_batch(tab.data_ptr(), len(jobs), b, None)
    torch.cuda.synchronize()
    for (w, taps, K, nc, st, sk, sn, flip), d1, d2 in zip(jobs, single, batched):
        ref, ws, amax = _pack_ref(w.cpu().numpy(), taps, K, nc, st, sk, sn, flip)
        for d in (d1, d2):
            raw = d.cpu().numpy()
            img = raw[:-4].view(np.float16)
            assert np.array_equal(img.view(np.uint16), ref.view(np.uint16))
            assert raw[-4] == ws and raw[-3] == amax


@pytest.mark.parametrize('parts,groups,C,chunks', [(1000, 1, 32, 128), (4097, 4, 32, 128), (130, 4, 512, 128), (16384, 4, 32, 128)])
def test_bn_stats_rows_fold(dclib, parts, groups, C, chunks):
    """dc_bn_stats_rows_fold: chunk b holds the sum of rows [parts*b/chunks, parts*(b+1)/chunks) (float64, to rounding), and the
    finalize over the folded rows gives the statistics of the finalize over all of them."""
    L = dclib
    rs = np.random.RandomState(parts + C)
    part = rs.standard_normal((parts, groups * C, 2))
    part[..., 1] = np.abs(part[..., 1]) * 3 + 2.0           # sums of squares: keep the variance positive
    p_d = dev(part)
    out = torch.zeros(chunks * groups * C * 2, dtype=torch.float64, device='cuda')
    L.dc_bn_stats_rows_fold(p_d.data_ptr(), parts, groups, C, chunks, out.data_ptr(), None)
    got = out.cpu().numpy().reshape(chunks, groups * C, 2)
    for b in (0, 1, chunks // 2, chunks - 1):
        r0, r1 = parts * b // chunks, parts * (b + 1) // chunks
        ref = part[r0:r1].sum(axis=0)
        assert np.allclose(got[b], ref, rtol=1e-12, atol=1e-11)
    assert np.allclose(got.sum(axis=0), part.sum(axis=0), rtol=1e-12, atol=1e-10)
    count = float(parts * groups * 7)
    res = []
    for src, n in ((p_d, parts), (out, chunks)):
        mean = torch.zeros(C, device='cuda')
        inv = torch.zeros(C, device='cuda')
        L.dc_bn_stats_finalize(src.data_ptr(), n, groups, C, count, 1e-3, -1.0, mean.data_ptr(), inv.data_ptr(), None, None, None)
        res.append((mean.cpu().numpy(), inv.cpu().numpy()))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-6, atol=1e-9) and np.allclose(res[0][1], res[1][1], rtol=1e-6)
    with pytest.raises(Exception):
        L.dc_bn_stats_rows_fold(p_d.data_ptr(), parts, groups, C, parts + 1, out.data_ptr(), None)
