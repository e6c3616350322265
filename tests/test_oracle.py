"""The oracle pinned: A (numpy float64, hand-derived backward) == B (torch float64 autograd) to 1e-10, finite
differences, analytic known-answer tests for every third-party semantic of SURVEY Appendix A, and the metric
algebra against reference-generated goldens (tests/test_host.py)."""
import numpy as np
import pytest
import torch

from oracle import unet_numpy as on
from oracle.unet_torch import UNetTorch


def test_oracle_a_equals_oracle_b_whole_net():
    nfb, N, H, W = 4, 2, 32, 32
    Wt = on.init_weights(nfb, randomize_bn=True, dtype=np.float64)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    A, B = on.UNetOracle(Wt, nfb), UNetTorch(Wt, nfb)
    la, pa, Ga, sa = A.loss_and_grads(x, y, masks)
    lb, pb, Gb, sb = B.loss_and_grads(x, y, masks)
    assert abs(la - lb) < 1e-12 and np.abs(pa - pb).max() < 1e-12
    for k in Ga:
        for a, b in zip(Ga[k], Gb[k]):
            assert np.abs(a.reshape(b.shape) - b).max() < 1e-10, k
    for k in sa:
        assert np.abs(sa[k][0] - sb[k][0]).max() < 1e-12 and np.abs(sa[k][1] - sb[k][1]).max() < 1e-12
    assert np.abs(A.forward(x) - B.forward(x).detach().numpy()).max() < 1e-12
    # two full train steps (Keras Adam + biased moving variance) stay together
    sA, sB = dict(it=0, m={}, v={}), dict(it=0, m={}, v={})
    for _ in range(2):
        l1, _ = A.train_step(x, y, sA, masks)
        l2, _ = B.train_step(x, y, sB, masks)
        assert abs(l1 - l2) < 1e-9
    for wa, wb in zip(A.weights(), [t.detach().numpy() for pl in B.P.values() for t in pl]):
        assert np.abs(wa - wb.reshape(wa.shape)).max() < 1e-7


def test_forced_gates_and_pool_indices():
    """UNetOracle / UNetTorch `force`: (i) forcing an oracle's OWN ReLU gates and pool indices changes nothing, bit for
    bit; (ii) with a few gates and indices flipped -- what an fp32 implementation does where a pre-activation or a
    window's top-two gap is within rounding of 0 -- the hand-derived backward (A) and autograd (B) still agree to
    1e-10, and the gradient really moves (so an un-forced comparison of such an implementation needs loose norms)."""
    nfb, N, H, W = 4, 2, 32, 32
    Wt = on.init_weights(nfb, randomize_bn=True, dtype=np.float64)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    A0 = on.UNetOracle(Wt, nfb)
    cache = {}
    A0.forward(x, True, masks, cache=cache)
    gates = {k: v[2].copy() for k, v in cache.items() if not k.startswith('p') and k != 'out'}
    pool = {lvl: cache['p%d' % lvl].copy() for lvl in range(4)}
    l0, p0, G0, _ = A0.loss_and_grads(x, y, masks)
    l1, p1, G1, _ = on.UNetOracle(Wt, nfb, force=dict(gates=gates, pool=pool)).loss_and_grads(x, y, masks)
    assert l0 == l1 and np.array_equal(p0, p1)
    for k in G0:
        for a, b in zip(G0[k], G1[k]):
            assert np.array_equal(a, b), k
    rs = np.random.RandomState(3)
    for k in ('e0b', 'd1a', 'u2'):
        flip = rs.random_sample(gates[k].shape) < 0.01
        gates[k] = gates[k] ^ flip
    pool[1] = ((pool[1] + (rs.random_sample(pool[1].shape) < 0.02)) % 4).astype(np.uint8)
    force = dict(gates=gates, pool=pool)
    la, pa, Ga, _ = on.UNetOracle(Wt, nfb, force=force).loss_and_grads(x, y, masks)
    lb, pb, Gb, _ = UNetTorch(Wt, nfb, force=force).loss_and_grads(x, y, masks)
    assert abs(la - lb) < 1e-12 and np.abs(pa - pb).max() < 1e-12
    moved = 0.0
    for k in Ga:
        for a, b, c in zip(Ga[k], Gb[k], G0[k]):
            assert np.abs(a.reshape(b.shape) - b).max() < 1e-10, k
            moved = max(moved, np.abs(a - c).max())
    assert moved > 1e-4
    # inference never forces
    assert np.array_equal(on.UNetOracle(Wt, nfb, force=force).forward(x), A0.forward(x))


def test_finite_difference_head_bias():
    """Catches sub-gradient artefacts at p == 0.5 exactly (relu/abs formulations of the BCE are off by 0.5 there)."""
    nfb, N, H, W = 4, 1, 16, 16
    Wt = on.init_weights(nfb, randomize_bn=True, dtype=np.float64)
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(nfb, N, H, W)
    _, _, G, _ = on.UNetOracle(Wt, nfb).loss_and_grads(x, y, masks)

    def loss(db):
        W2 = [w.copy() for w in Wt]
        W2[-1] = W2[-1] + db
        o = on.UNetOracle(W2, nfb)
        return on.bce_keras(o.forward(x, True, masks), y.astype(float))
    e = 1e-6
    fd = [(loss(np.eye(2)[i] * e) - loss(-np.eye(2)[i] * e)) / (2 * e) for i in range(2)]
    assert np.allclose(fd, G['out'][1], atol=1e-8)


def test_conv_semantics_kat():
    """'same' padding, cross-correlation (no flip), HWIO layout."""
    x = np.zeros((1, 5, 5, 1))
    x[0, 2, 2, 0] = 1.0
    K = np.arange(9, dtype=float).reshape(3, 3, 1, 1)
    z = on.conv3x3_fwd(x, K, np.zeros(1))[0, :, :, 0]
    # cross-correlation of a delta reproduces the FLIPPED kernel around the delta
    assert np.array_equal(z[1:4, 1:4], K[::-1, ::-1, 0, 0])
    # torch agrees
    zt = torch.nn.functional.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(K).permute(3, 2, 0, 1), padding=1)
    assert np.array_equal(zt[0, 0].numpy(), z)


def test_convT_semantics_kat():
    """out[n,2i+a,2j+b,o] = b[o] + sum_c x[n,i,j,c] K[a,b,o,c]; kernel layout (2,2,Cout,Cin)."""
    rs = np.random.RandomState(0)
    x, K, b = rs.standard_normal((1, 3, 4, 5)), rs.standard_normal((2, 2, 7, 5)), rs.standard_normal(7)
    z = on.convT2x2_fwd(x, K, b)
    assert z.shape == (1, 6, 8, 7)
    assert np.allclose(z[0, 2 * 1 + 1, 2 * 2 + 0], b + K[1, 0] @ x[0, 1, 2])
    zt = torch.nn.functional.conv_transpose2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(K).permute(3, 2, 0, 1),
                                              torch.tensor(b), stride=2)
    assert np.allclose(zt.permute(0, 2, 3, 1).numpy(), z)


def test_batchnorm_semantics_kat():
    rs = np.random.RandomState(1)
    z = rs.standard_normal((2, 4, 4, 3)) * 3 + 1
    y, (xh, inv, mu, var) = on.bn_train_fwd(z, np.ones(3), np.zeros(3))
    assert np.allclose(var, z.reshape(-1, 3).var(0, ddof=0))            # biased / population variance
    assert np.allclose(inv, 1 / np.sqrt(var + 1e-3))                     # eps = 1e-3
    const = np.full((2, 4, 4, 3), 2.5)
    y, _ = on.bn_train_fwd(const, np.ones(3), np.array([0.1, -0.2, 0.3]))
    assert np.allclose(y, [0.1, -0.2, 0.3])                              # var = 0 => output = beta


def test_maxpool_first_max_tiebreak():
    x = np.zeros((1, 2, 4, 1))
    x[0, :, :2, 0] = 1.0                   # all-equal window -> index 0
    x[0, 0, 3, 0] = x[0, 1, 2, 0] = 5.0    # tie between positions 1 and 2 -> 1 (row-major first)
    p, idx = on.maxpool2x2_fwd(x)
    assert idx[0, 0, 0, 0] == 0 and idx[0, 0, 1, 0] == 1
    dx = on.maxpool2x2_bwd(np.ones_like(p), idx)
    assert dx[0, 0, 0, 0] == 1 and dx[0, 0, 3, 0] == 1 and dx.sum() == 2
    # torch CPU max_pool2d breaks ties the same way
    pt, it = torch.nn.functional.max_pool2d(torch.tensor(x).permute(0, 3, 1, 2), 2, 2, return_indices=True)
    assert it[0, 0, 0, 1].item() == 0 * 4 + 3


def test_bce_clip_and_adam_closed_forms():
    # float32 clip bounds of the reference graph: p beyond them contributes the clipped value and zero gradient
    p = np.array([1e-9, 0.5, 1 - 1e-9])
    y = np.array([1.0, 1.0, 0.0])
    lo, hi = on.CLIP_LO, on.CLIP_HI
    assert hi == float(np.float32(1) - np.float32(1e-7)) and hi < 1 - 1e-7
    exp = np.mean([-np.log(lo), -np.log(0.5), -np.log(1 - hi)])
    assert abs(on.bce_keras(p, y) - exp) < 1e-9
    g = on.bce_keras_grad(p, y)
    assert g[0] == 0 and g[2] == 0 and abs(g[1] - (0.5 - 1) / 0.25 / 3) < 1e-12
    # Keras Adam step 1 on g = 1: delta = -lr_t * (1-b1) / (sqrt(1-b2) + eps), lr_t = lr*sqrt(1-b2)/(1-b1)
    pnew, m, v = on.adam_keras(np.zeros(1), np.ones(1), np.zeros(1), np.zeros(1), 0, lr=0.002)
    lr_t = 0.002 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(pnew[0] + lr_t * 0.1 / (np.sqrt(0.001) + 1e-8)) < 1e-15
    assert abs(pnew[0] + 0.002) < 1e-8           # ~ -lr, but NOT exactly: eps sits outside the bias correction
    assert pnew[0] != -0.002


def test_weight_list_layout():
    shapes = on.weight_shapes(32)
    assert len(shapes) == 134
    assert sum(int(np.prod(s)) for s in shapes) == 7773250
    assert shapes[0] == (3, 3, 1, 32) and shapes[-2] == (1, 1, 32, 2)
    t = on.layer_table(32)
    up = [l for l in t if l[1] == 'convT']
    assert [(l[2], l[3], l[4]) for l in up] == [(512, 256, 0.5), (256, 128, 0.5), (128, 64, 0.5), (64, 32, 0.5)]
    assert on.dropout_rates(0.25) == {'e1b': 0.25, 'e2b': 0.5, 'e3b': 0.5, 'u3': 0.5, 'u2': 0.5, 'u1': 0.5, 'u0': 0.25}
