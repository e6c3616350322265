"""GPU parity of the "dz on load" BatchNorm backward (include/dcunet.h: dc_bn_bwd_finalize_dzin,
dc_conv3x3_dgrad_dzin_f16x3, dc_conv3x3_wgrad_dzin_f16x3) against the float64 numpy oracle, through the C ABI.

Reference ops: Conv2D -> BatchNormalization -> Activation('relu') of a Dropout-free block,
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:163-167 (instances :172-173, :219-220): the gradient of
the block w.r.t. its pre-BN tensor is  dz = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)),  dy = da*[relu gate]
(oracle/unet_numpy.py bn_train_bwd).  The device never materialises dz: both gradient kernels form it from (da, z) and
the per-channel table of dc_bn_bwd_finalize_dzin.  Tolerance: 2e-5 of the output scale, as for the other contractions."""
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


def _block_case(rs, N, H, W, Cin, Cout, big_mean=False):
    """Random Dropout-free block: x (input), z (pre-BN output), batch statistics of z, da.  Returns float32 arrays."""
    z = rs.standard_normal((N, H, W, Cout)).astype(np.float32) * (rs.random_sample(Cout).astype(np.float32) + 0.5)
    if big_mean:
        z += 50.0
    z64 = z.reshape(-1, Cout).astype(np.float64)
    mean = z64.mean(0).astype(np.float32)
    invstd = (1.0 / np.sqrt(z64.var(0) + 1e-3)).astype(np.float32)
    gamma = (rs.standard_normal(Cout) * 0.7 + 1.0).astype(np.float32)
    beta = (rs.standard_normal(Cout) * 0.4).astype(np.float32)
    da = (rs.standard_normal((N, H, W, Cout)) * 3e-3).astype(np.float32)
    x = rs.standard_normal((N, H, W, Cin)).astype(np.float32)
    return x, z, mean, invstd, gamma, beta, da


def _dz_ref(z, mean, invstd, gamma, beta, da):
    """float64 dz with the DEVICE's ReLU gate (sign of the exact fmaf(z, sc, sh) on fp32 sc / sh: dc_bn_affine)."""
    C = z.shape[-1]
    sc = (gamma * invstd).astype(np.float32)
    sh = (beta.astype(np.float64) - mean.astype(np.float64) * sc.astype(np.float64)).astype(np.float32)
    z64, da64 = z.reshape(-1, C).astype(np.float64), da.reshape(-1, C).astype(np.float64)
    gate = (z64 * sc.astype(np.float64) + sh.astype(np.float64)) > 0
    dy = np.where(gate, da64, 0.0)
    xh = (z64 - mean.astype(np.float64)) * invstd.astype(np.float64)
    M = z64.shape[0]
    dbeta, dgamma = dy.sum(0), (dy * xh).sum(0)
    dz = gamma.astype(np.float64) * invstd.astype(np.float64) * (dy - dbeta / M - xh * dgamma / M)
    return dz.reshape(z.shape), dy.reshape(z.shape), dgamma, dbeta


def _finalize(L, z, mean, invstd, gamma, beta, da, count=None):
    """dc_bn_bwd_reduce (sums + max |dy|) -> dc_bn_bwd_finalize_dzin.  Returns the device tensors the consumers take."""
    N, H, W, C = z.shape
    M = N * H * W
    zd, dad = dev(z), dev(da)
    md, isd, gd, bd = dev(mean), dev(invstd), dev(gamma), dev(beta)
    if _pow2_c(C):
        blocks = L.dc_bn_bwd_blocks(M, C)
        part = torch.full((blocks * C * 2,), float('nan'), device='cuda')
        amx = torch.full((blocks * C,), float('nan'), device='cuda')
        L.dc_bn_bwd_reduce(dad.data_ptr(), C, zd.data_ptr(), md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), None, 1.0, 0,
                           part.data_ptr(), amx.data_ptr(), M, C, None)
    else:
        # dc_bn_bwd_reduce takes power-of-two channel counts; the finalize kernel and the consumers do not: pass-1 partial rows
        # (sum dy, sum dy*xhat) and max |dy| built on the host in float64, 3 rows of pixels (the product's producers for such
        # layers would be the kernels that write da)
        _, dy, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
        dy = dy.reshape(-1, C)
        xh = (z.reshape(-1, C).astype(np.float64) - mean.astype(np.float64)) * invstd.astype(np.float64)
        blocks = 3
        cuts = [0, M // 3, 2 * M // 3, M]
        rows = [np.stack([dy[a:b].sum(0), (dy[a:b] * xh[a:b]).sum(0)], axis=1) for a, b in zip(cuts[:-1], cuts[1:])]
        part = dev(np.stack(rows).astype(np.float32).reshape(-1))
        amx = dev(np.stack([np.abs(dy[a:b]).max(0) for a, b in zip(cuts[:-1], cuts[1:])]).astype(np.float32).reshape(-1))
    dg, db = torch.full((C,), float('nan'), device='cuda'), torch.full((C,), float('nan'), device='cuda')
    coef = torch.full((7 * C,), float('nan'), device='cuda')
    dbias = torch.full((C,), float('nan'), device='cuda')
    L.dc_bn_bwd_finalize_dzin(part.data_ptr(), amx.data_ptr(), blocks, C, md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                              float(count or M), dg.data_ptr(), db.data_ptr(), coef.data_ptr(), dbias.data_ptr(), None)
    _KEEP.extend([part, amx, dg, db, coef, dbias])
    return zd, dad, coef, dg, db, dbias


def _pow2_c(C):
    return C & (C - 1) == 0


@pytest.mark.parametrize('N,H,W,C', [(2, 16, 16, 32), (1, 24, 40, 64), (3, 8, 8, 8), (2, 32, 32, 256)])
def test_finalize_dzin_table_and_bound(dclib, N, H, W, C):
    """The table reproduces dz (float64) from (da, z) with the kernels' own expression, the bound dominates max |dz_c| --
    and is not absurdly loose (the fp16 split keeps 22 bits down to 2^-12 of it) --, dgamma / dbeta equal
    dc_bn_bwd_finalize's, the conv-bias gradient slot is an exact 0."""
    L = dclib
    rs = np.random.RandomState(C + H)
    _, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, 4, C)
    dz_ref, dy_ref, dg_ref, db_ref = _dz_ref(z, mean, invstd, gamma, beta, da)
    zd, dad, coef, dg, db, dbias = _finalize(L, z, mean, invstd, gamma, beta, da)
    torch.cuda.synchronize()
    t = coef.cpu().numpy().reshape(7, C).astype(np.float64)
    assert np.abs(dg.cpu().numpy() - dg_ref).max() < 2e-5 * max(np.abs(dg_ref).max(), 1e-30)
    assert np.abs(db.cpu().numpy() - db_ref).max() < 2e-5 * max(np.abs(db_ref).max(), 1e-30)
    assert np.all(dbias.cpu().numpy() == 0.0)
    z64 = z.reshape(-1, C).astype(np.float64)
    y = z64 * t[0] + t[1]
    dy = np.where(y > 0, da.reshape(-1, C).astype(np.float64), 0.0)
    dz_tab = t[3] * dy + (t[4] * (z64 - t[2]) + t[5])
    scale = np.abs(dz_ref).max()
    assert np.abs(dz_tab - dz_ref.reshape(-1, C)).max() < 2e-5 * scale
    amax_dz = np.abs(dz_ref).reshape(-1, C).max(0)
    assert np.all(t[6] >= amax_dz * (1 - 1e-6))
    assert t[6].max() <= 64 * scale, (t[6].max(), scale)


# N, H, W, Cin, Cout: shapes the role-split kernel serves as a data gradient (W > 16, Cout a multiple of 16 and >= 32);
# both tile shapes (Cin <= 32: 16 x 32 pixels x 32 columns; else 8 x 32 x 64), ragged tiles, 2 .. 16 steps per tile
DGRAD_SHAPES = [(2, 64, 64, 64, 64), (1, 40, 72, 48, 80), (2, 48, 48, 32, 32), (1, 32, 32, 128, 256), (3, 33, 50, 64, 96),
                (1, 64, 96, 64, 32), (2, 32, 64, 32, 64)]


@pytest.mark.parametrize('N,H,W,Cin,Cout', DGRAD_SHAPES)
@pytest.mark.parametrize('with_sums', [False, True])
def test_conv3x3_dgrad_dzin(dclib, N, H, W, Cin, Cout, with_sums):
    """dx = conv3x3_transpose(dz) with dz formed on load, against the float64 oracle (conv3x3_bwd on the float64 dz), and --
    with_sums -- the pass-1 sums / max |dy| it emits for the layer in front against a float64 reduction of the dx it wrote."""
    L = dclib
    rows = L.dc_conv3x3_dgrad_dzin_blocks(N, H, W, Cin, Cout)
    assert rows > 0
    rs = np.random.RandomState(Cin + Cout + H)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    dx_ref, _, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    Kd = dev(K)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    dx = torch.full((N, H, W, Cin), float('nan'), device='cuda')
    red = (None,) * 7
    if with_sums:
        rz = dev(rs.standard_normal((N, H, W, Cin)).astype(np.float32))
        rmu = dev((rs.standard_normal(Cin) * 0.2).astype(np.float32)); ris = dev((rs.random_sample(Cin) + 0.5).astype(np.float32))
        rga = dev(rs.standard_normal(Cin).astype(np.float32)); rbe = dev((rs.standard_normal(Cin) * 0.3).astype(np.float32))
        part = torch.full((rows * Cin * 2,), float('nan'), device='cuda')
        amx = torch.full((rows * Cin,), float('nan'), device='cuda')
        red = (rz.data_ptr(), rmu.data_ptr(), ris.data_ptr(), rga.data_ptr(), rbe.data_ptr(), part.data_ptr(), amx.data_ptr())
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, *red,
                                  N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    got = dx.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - dx_ref).max() < 2e-5 * np.abs(dx_ref).max()
    # run to run: bit-reproducible
    dx2 = torch.full_like(dx, float('nan'))
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx2.data_ptr(), None, *red,
                                  N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    if with_sums:
        dg, db = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
        L.dc_bn_bwd_finalize(part.data_ptr(), rows, Cin, dg.data_ptr(), db.data_ptr(), None)
        torch.cuda.synchronize()
        sc = (rga * ris).cpu().numpy()
        sh = (rbe.cpu().numpy().astype(np.float64) - rmu.cpu().numpy().astype(np.float64) * sc.astype(np.float64)).astype(np.float32)
        za = rz.cpu().numpy().astype(np.float64).reshape(-1, Cin)
        gate = (za * sc.astype(np.float64) + sh.astype(np.float64)) > 0
        dy = np.where(gate, got.astype(np.float64).reshape(-1, Cin), 0.0)
        ref_db = dy.sum(0)
        ref_dg = (dy * (za - rmu.cpu().numpy().astype(np.float64)) * ris.cpu().numpy().astype(np.float64)).sum(0)
        tol = 2e-5 * max(np.abs(ref_dg).max(), np.abs(ref_db).max())
        assert np.abs(dg.cpu().numpy() - ref_dg).max() < tol and np.abs(db.cpu().numpy() - ref_db).max() < tol
        assert np.array_equal(amx.cpu().numpy().reshape(rows, Cin).max(0), np.abs(dy).max(0).astype(np.float32))


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 64, 64, 64, 64), (1, 40, 72, 48, 80), (1, 32, 32, 128, 256), (3, 33, 50, 64, 96),
                                            (2, 32, 64, 256, 128)])
def test_dgrad_dzin_writes_dz_for_a_plain_weight_gradient(dclib, N, H, W, Cin, Cout):
    """dz_out (round 5): the dz-on-load data gradient also WRITES the dz its producers form -- every pixel exactly once (interior
    and ragged tiles, 2-5 column blocks of which only the first stores) -- so that the block's weight gradient is the plain kernel
    and no BatchNorm-backward apply pass runs.  dx is bit-identical with and without the write-back; dz_out is the float64 dz to
    fp32 rounding; dc_conv3x3_wgrad_f16x3 on it (scale from the table's bound row) meets the oracle's dK at the usual 2e-5."""
    L = dclib
    assert L.dc_conv3x3_dgrad_dzin_blocks(N, H, W, Cin, Cout) > 0
    rs = np.random.RandomState(Cin + Cout + W)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    _, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    Kd, xd = dev(K), dev(x)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    dx0 = torch.full((N, H, W, Cin), float('nan'), device='cuda'); dx1 = torch.full_like(dx0, float('nan'))
    dzo = torch.full((N, H, W, Cout), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx0.data_ptr(), None, *((None,) * 7),
                                  N, H, W, Cin, Cout, None)
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx1.data_ptr(), dzo.data_ptr(),
                                  *((None,) * 7), N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    got = dzo.cpu().numpy()
    assert np.isfinite(got).all(), 'pixels the write-back missed: %d' % int((~np.isfinite(got)).sum())
    assert np.abs(got - dz_ref).max() < 2e-6 * np.abs(dz_ref).max()
    scale = torch.full((4,), float('nan'), device='cuda')
    L.dc_pow2_scale_from_absmax(coef.data_ptr() + 4 * 6 * Cout, Cout, 1024.0, scale.data_ptr(), None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), device='cuda')
    dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_f16x3(xd.data_ptr(), dzo.data_ptr(), dw.data_ptr(), ws.data_ptr(), scale.data_ptr(), None, N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    sc = float(scale[0].item())
    assert sc > 0 and np.log2(sc) == np.floor(np.log2(sc)) and sc * np.abs(got).max() <= 1024.0
    assert np.abs(dw.cpu().numpy() - dK_ref).max() < 2e-5 * np.abs(dK_ref).max()
    if Cin <= 32:
        return
    from deep_calcium_amd._lib import DcunetError
    with pytest.raises(DcunetError, match='dz_out needs more than 32|shape not served'):
        small = torch.empty((N, H, W, 32), device='cuda')
        wps = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, 32), device='cuda')
        L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wps.data_ptr(), small.data_ptr(), dzo.data_ptr(),
                                      *((None,) * 7), N, H, W, 32, Cout, None)


# every weight-gradient tile configuration (wgrad_f16x3.hip CONV_H_DISPATCH) incl. the narrow ones the data gradient
# leaves to the apply path, ragged tiles, x materialised / BN + ReLU on load
WGRAD_SHAPES = [(2, 64, 64, 64, 64), (1, 40, 72, 48, 32), (2, 48, 48, 32, 32), (1, 32, 32, 128, 256), (2, 16, 16, 64, 128),
                (2, 8, 8, 128, 64), (1, 64, 96, 64, 32), (2, 32, 64, 32, 64), (3, 33, 50, 64, 64)]


@pytest.mark.parametrize('N,H,W,Cin,Cout', WGRAD_SHAPES)
@pytest.mark.parametrize('bnin', [False, True])
def test_conv3x3_wgrad_dzin(dclib, N, H, W, Cin, Cout, bnin):
    L = dclib
    rs = np.random.RandomState(Cin * 3 + Cout + W)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    xs, xsh = None, None
    x_eff = x.astype(np.float64)
    if bnin:
        xsc = (rs.random_sample(Cin) + 0.5).astype(np.float32); xshv = (rs.standard_normal(Cin) * 0.3).astype(np.float32)
        x_eff = np.maximum(x.astype(np.float64) * xsc.astype(np.float64) + xshv.astype(np.float64), 0.0)
        xs, xsh = dev(xsc), dev(xshv)
    K = np.zeros((3, 3, Cin, Cout))
    _, dK_ref, _ = on.conv3x3_bwd(x_eff, K, dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    xd = dev(x)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), device='cuda')
    dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
    args = (xd.data_ptr(), xs.data_ptr() if bnin else None, xsh.data_ptr() if bnin else None, None, dad.data_ptr(), zd.data_ptr(),
            coef.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, Cin, Cout, None)
    L.dc_conv3x3_wgrad_dzin_f16x3(*args)
    torch.cuda.synchronize()
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - dK_ref).max() < 2e-5 * np.abs(dK_ref).max()
    dw2 = torch.full_like(dw, float('nan'))
    L.dc_conv3x3_wgrad_dzin_f16x3(*(args[:7] + (dw2.data_ptr(),) + args[8:]))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('N,H,W,Cout', [(2, 32, 32, 32), (1, 17, 23, 8), (2, 64, 64, 16)])
def test_first_layer_wgrad_dzin(dclib, N, H, W, Cout):
    """Cin == 1: x is the (N,H,W) image (conv_c1.hip; W % 4 != 0 takes the generic kernel)."""
    L = dclib
    rs = np.random.RandomState(W + Cout)
    _, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, 4, Cout)
    x = rs.standard_normal((N, H, W, 1)).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    _, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), np.zeros((3, 3, 1, Cout)), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    xd = dev(x.reshape(N, H, W))
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, 1, Cout), device='cuda')
    dw = torch.full((3, 3, 1, Cout), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_dzin_f16x3(xd.data_ptr(), None, None, None, dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), dw.data_ptr(),
                                  ws.data_ptr(), N, H, W, 1, Cout, None)
    torch.cuda.synchronize()
    assert np.abs(dw.cpu().numpy() - dK_ref).max() < 2e-5 * np.abs(dK_ref).max()


@pytest.mark.parametrize('case', ['tiny', 'huge', 'mean50', 'zero'])
def test_dzin_fp16_range(dclib, case):
    """The bound-derived power of two keeps the split-fp16 operands in range whatever the gradient's magnitude: da of 1e-12
    and of 1e6, a pre-BN tensor at |mean| = 50 sigma (z - mu is formed before the multiply: no cancellation against E), and
    an all-zero gradient (scale 1, exact zeros out)."""
    L = dclib
    N, H, W, Cin, Cout = 2, 32, 64, 64, 64
    rs = np.random.RandomState(11)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout, big_mean=(case == 'mean50'))
    da = da * {'tiny': 1e-9, 'huge': 3e8, 'mean50': 1.0, 'zero': 0.0}[case]
    da = da.astype(np.float32)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    dx_ref, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    Kd, xd = dev(K), dev(x)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    dx = torch.full((N, H, W, Cin), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, *((None,) * 7),
                                  N, H, W, Cin, Cout, None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), device='cuda')
    dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_dzin_f16x3(xd.data_ptr(), None, None, None, dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), dw.data_ptr(),
                                  ws.data_ptr(), N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    gx, gw = dx.cpu().numpy(), dw.cpu().numpy()
    assert np.isfinite(gx).all() and np.isfinite(gw).all()
    if case == 'zero':
        assert np.all(gx == 0) and np.all(gw == 0)
    else:
        tol = 1e-4 if case == 'mean50' else 2e-5
        assert np.abs(gx - dx_ref).max() < tol * np.abs(dx_ref).max()
        assert np.abs(gw - dK_ref).max() < tol * np.abs(dK_ref).max()


@pytest.mark.parametrize('noise', [1e-2, 1e-3])
def test_dzin_bound_when_dy_follows_xhat(dclib, noise):
    """The dz-on-load bound adds the worst-case term |dgamma| sqrt(M) / M.  When the incoming gradient follows xhat
    (da = s xhat + noise), BatchNorm's backward projects that component OUT: the true max |dz| is ~noise, the bound ~sqrt(M) s
    -- 2^13 .. 2^17 above it at M = 2^18 -- and the power-of-two scale derived from the bound parks dz low in fp16's range.
    The split still keeps hi + lo to an ABSOLUTE 2^-24 of the scaled value (fp16 subnormal spacing), i.e. 2^-22 relative to
    the scaled maximum at a 2^17-fold loose bound: the contractions stay inside the 2e-5 bound of every other test.  (They
    would leave it near a 2^24-fold loose bound: noise 1e-5 s, below fp32's own rounding of da.)"""
    L = dclib
    N, H, W, Cin, Cout = 4, 256, 256, 32, 32
    M = N * H * W
    rs = np.random.RandomState(5)
    x, z, mean, invstd, gamma, beta, _ = _block_case(rs, N, H, W, Cin, Cout)
    gamma = (np.abs(gamma) * 0.3 + 0.5).astype(np.float32)
    beta = (np.abs(beta) + 10.0).astype(np.float32)   # y = gamma xhat + beta > 0 for every |xhat| < 6: all gates open, dy = da
    xh = (z.astype(np.float64) - mean.astype(np.float64)) * invstd.astype(np.float64)
    s_c = 3e-3 * (rs.random_sample(Cout) + 0.5)
    da = (s_c * (xh + noise * rs.standard_normal(z.shape))).astype(np.float32)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    torch.cuda.synchronize()
    bound = coef.cpu().numpy().reshape(7, Cout)[6].max()
    loose = bound / np.abs(dz_ref).max()
    assert loose > 2.0 ** 12, loose                   # the case the advisor asked for: the bound sits far above max |dz|
    dx_ref, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz_ref)
    Kd, xd = dev(K), dev(x)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    dx = torch.full((N, H, W, Cin), float('nan'), device='cuda')
    L.dc_conv3x3_dgrad_dzin_f16x3(dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx.data_ptr(), None, *((None,) * 7),
                                  N, H, W, Cin, Cout, None)
    ws = torch.empty(L.dc_conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout), device='cuda')
    dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
    L.dc_conv3x3_wgrad_dzin_f16x3(xd.data_ptr(), None, None, None, dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), dw.data_ptr(),
                                  ws.data_ptr(), N, H, W, Cin, Cout, None)
    torch.cuda.synchronize()
    ex = np.abs(dx.cpu().numpy() - dx_ref).max() / np.abs(dx_ref).max()
    ew = np.abs(dw.cpu().numpy() - dK_ref).max() / np.abs(dK_ref).max()
    print('dy ~ xhat, noise %g: bound / max|dz| = 2^%.1f, dx error %.2e, dW error %.2e' % (noise, np.log2(loose), ex, ew))
    # forming dz in fp32 cancels A dy (~s xhat) against D (z - mu) (~ -s xhat): each term rounds at 6e-8 s, what is left is
    # ~noise s -- a floor of ~6e-8 / noise per element for ANY fp32 evaluation (the apply pass included), averaged down by
    # the contraction
    tol = 2e-5 + 4 * 6e-8 / noise
    assert ex < tol and ew < tol, (ex, ew)


@pytest.mark.parametrize('N,H,W', [(2, 64, 64), (1, 40, 72), (3, 33, 50), (1, 128, 96), (16, 32, 32)])
@pytest.mark.parametrize('bnin', [True, False])
def test_conv3x3_bwd_joint(dclib, N, H, W, bnin):
    """dc_conv3x3_bwd_joint_f16x3 (Cin = Cout = 32): dx, dW and the pass-1 sums of the layer in front from ONE kernel, against
    the float64 oracle on the float64 dz -- interior and ragged tiles (H % 4, W % 32), more tiles than workgroups and fewer,
    x materialised / BN + ReLU on load; bit-reproducible run to run."""
    L = dclib
    Cin = Cout = 32
    rows = L.dc_conv3x3_bwd_joint_blocks(N, H, W, Cin, Cout)
    assert rows > 0 and rows % 2 == 0
    assert L.dc_conv3x3_bwd_joint_blocks(N, H, W, 64, 64) == 0 and L.dc_conv3x3_bwd_joint_blocks(N, H, 16, 32, 32) == 0
    rs = np.random.RandomState(H * 7 + W)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    # the layer in front: x is ITS pre-BN tensor when bnin (the sums are about that layer), a materialised activation otherwise
    rmu = (rs.standard_normal(Cin) * 0.2).astype(np.float32); ris = (rs.random_sample(Cin) + 0.5).astype(np.float32)
    rga = (rs.standard_normal(Cin) * 0.5 + 1.0).astype(np.float32); rbe = (rs.standard_normal(Cin) * 0.3).astype(np.float32)
    xsc = (rga * ris).astype(np.float32)
    xsh = (rbe.astype(np.float64) - rmu.astype(np.float64) * xsc.astype(np.float64)).astype(np.float32)
    x_eff = np.maximum(x.astype(np.float64) * xsc.astype(np.float64) + xsh.astype(np.float64), 0.0) if bnin else x.astype(np.float64)
    dx_ref, dK_ref, _ = on.conv3x3_bwd(x_eff, K.astype(np.float64), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    Kd, xd = dev(K), dev(x)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    ws = torch.empty(L.dc_conv3x3_bwd_joint_ws_floats(N, H, W, Cin, Cout), device='cuda')
    rmud, risd, rgad, rbed, xscd, xshd = dev(rmu), dev(ris), dev(rga), dev(rbe), dev(xsc), dev(xsh)

    def run():
        dx = torch.full((N, H, W, Cin), float('nan'), device='cuda')
        dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
        part = torch.full((rows * Cin * 2,), float('nan'), device='cuda')
        amx = torch.full((rows * Cin,), float('nan'), device='cuda')
        L.dc_conv3x3_bwd_joint_f16x3(xd.data_ptr(), xscd.data_ptr() if bnin else None, xshd.data_ptr() if bnin else None, None,
                                     dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(), dx.data_ptr(),
                                     xd.data_ptr(), rmud.data_ptr(), risd.data_ptr(), rgad.data_ptr(), rbed.data_ptr(),
                                     part.data_ptr(), amx.data_ptr(), dw.data_ptr(), ws.data_ptr(), N, H, W, Cin, Cout, None)
        torch.cuda.synchronize()
        return dx, dw, part, amx

    dx, dw, part, amx = run()
    gx, gw = dx.cpu().numpy(), dw.cpu().numpy()
    assert np.isfinite(gx).all() and np.isfinite(gw).all()
    assert np.abs(gx - dx_ref).max() < 2e-5 * np.abs(dx_ref).max(), np.abs(gx - dx_ref).max() / np.abs(dx_ref).max()
    assert np.abs(gw - dK_ref).max() < 2e-5 * np.abs(dK_ref).max(), np.abs(gw - dK_ref).max() / np.abs(dK_ref).max()
    dx2, dw2, part2, amx2 = run()
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(part, part2) and torch.equal(amx, amx2)
    # pass-1 sums of the layer in front (pre-BN tensor x, statistics rmu / ris, affine rga / rbe) from the dx it wrote
    dg, db = torch.zeros(Cin, device='cuda'), torch.zeros(Cin, device='cuda')
    L.dc_bn_bwd_finalize(part.data_ptr(), rows, Cin, dg.data_ptr(), db.data_ptr(), None)
    torch.cuda.synchronize()
    za = x.astype(np.float64).reshape(-1, Cin)
    gate = (za * xsc.astype(np.float64) + xsh.astype(np.float64)) > 0
    dy = np.where(gate, gx.astype(np.float64).reshape(-1, Cin), 0.0)
    ref_db, ref_dg = dy.sum(0), (dy * (za - rmu.astype(np.float64)) * ris.astype(np.float64)).sum(0)
    tol = 2e-5 * max(np.abs(ref_dg).max(), np.abs(ref_db).max())
    assert np.abs(dg.cpu().numpy() - ref_dg).max() < tol and np.abs(db.cpu().numpy() - ref_db).max() < tol
    assert np.array_equal(amx.cpu().numpy().reshape(rows, Cin).max(0), np.abs(dy).max(0).astype(np.float32))


@pytest.mark.parametrize('N,H,W', [(2, 64, 64), (1, 40, 72), (3, 33, 50), (16, 32, 32)])
def test_conv3x3_bwd_joint_64_to_32(dclib, N, H, W):
    """The 64 -> 32 variant (the first decoder convolution at full resolution: x = the concat buffer, materialised; no fused
    sums): dx (64 channels) and dW (3,3,64,32) against the float64 oracle, ragged tiles, bit-reproducible."""
    L = dclib
    Cin, Cout = 64, 32
    rows = L.dc_conv3x3_bwd_joint_blocks(N, H, W, Cin, Cout)
    assert rows > 0
    rs = np.random.RandomState(H * 3 + W)
    x, z, mean, invstd, gamma, beta, da = _block_case(rs, N, H, W, Cin, Cout)
    K = (rs.standard_normal((3, 3, Cin, Cout)) * 0.05).astype(np.float32)
    dz_ref, _, _, _ = _dz_ref(z, mean, invstd, gamma, beta, da)
    dx_ref, dK_ref, _ = on.conv3x3_bwd(x.astype(np.float64), K.astype(np.float64), dz_ref)
    zd, dad, coef, _, _, _ = _finalize(L, z, mean, invstd, gamma, beta, da)
    Kd, xd = dev(K), dev(x)
    wpd = torch.empty(L.dc_pack_weights_f16x3_floats(9, Cout, Cin), device='cuda')
    L.dc_pack_weights_f16x3(Kd.data_ptr(), wpd.data_ptr(), 9, Cout, Cin, Cin * Cout, 1, Cout, 1, None)
    ws = torch.empty(L.dc_conv3x3_bwd_joint_ws_floats(N, H, W, Cin, Cout), device='cuda')

    def run():
        dx = torch.full((N, H, W, Cin), float('nan'), device='cuda')
        dw = torch.full((3, 3, Cin, Cout), float('nan'), device='cuda')
        L.dc_conv3x3_bwd_joint_f16x3(xd.data_ptr(), None, None, None, dad.data_ptr(), zd.data_ptr(), coef.data_ptr(), wpd.data_ptr(),
                                     dx.data_ptr(), *((None,) * 7), dw.data_ptr(), ws.data_ptr(), N, H, W, Cin, Cout, None)
        torch.cuda.synchronize()
        return dx, dw

    dx, dw = run()
    gx, gw = dx.cpu().numpy(), dw.cpu().numpy()
    assert np.isfinite(gx).all() and np.isfinite(gw).all()
    assert np.abs(gx - dx_ref).max() < 2e-5 * np.abs(dx_ref).max(), np.abs(gx - dx_ref).max() / np.abs(dx_ref).max()
    assert np.abs(gw - dK_ref).max() < 2e-5 * np.abs(dK_ref).max(), np.abs(gw - dK_ref).max() / np.abs(dK_ref).max()
    dx2, dw2 = run()
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2)
    with pytest.raises(Exception):          # BN-on-load / fused sums are the 32 -> 32 kernel's
        L.dc_conv3x3_bwd_joint_f16x3(xd.data_ptr(), zd.data_ptr(), zd.data_ptr(), None, dad.data_ptr(), zd.data_ptr(), coef.data_ptr(),
                                     wpd.data_ptr(), dx.data_ptr(), *((None,) * 7), dw.data_ptr(), ws.data_ptr(), N, H, W, Cin, Cout, None)
