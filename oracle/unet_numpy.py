"""Oracle A (authoritative): float64 numpy restatement of the UNet2DS arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the
reference for these ops: the reference delegates them to Keras 2.0.6 / TF 1.2.1
(/root/reference/requirements.txt:28,:67, un-vendored).  Topology, ordering and
hyper-parameters follow /root/reference/deepcalcium/models/neurons/
unet_2d_summary.py:123-224; op semantics follow SURVEY.md Appendix A.

Layouts: activations NHWC, conv kernels HWIO (3,3,Cin,Cout), transposed-conv
kernels (2,2,Cout,Cin) -- the Keras `channels_last` conventions.
Every backward pass is hand-derived here; tests/test_oracle.py checks each one
against torch float64 autograd (oracle B) to 1e-10.
"""
import numpy as np

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon [3P]
K_EPS = 1e-7           # keras.backend.epsilon() [3P]
# The reference graph is float32: K.binary_crossentropy casts epsilon to the tensor dtype and forms
# `1 - eps` IN float32 (= 0.99999988..., not 1 - 1e-7).  The float64 oracle uses those float32 clip bounds,
# otherwise saturated pixels (p -> 1) are clipped by the reference but not by the oracle.
CLIP_LO = float(np.float32(1e-7))
CLIP_HI = float(np.float32(1.0) - np.float32(1e-7))


# ----------------------------------------------------------------------------
# Topology  (unet_2d_summary.py:169-223)
# ----------------------------------------------------------------------------
def layer_table(nfb=32, upsampling=False):
    """Weighted layers in graph-creation (= Keras get_weights) order.

    Returns a list of (name, kind, cin, cout, bn_momentum) with kind in
    {'conv', 'convT', 'head'}.  unet_2d_summary.py:172-221.  upsampling=True is the
    `upsampling_or_transpose != 'transpose'` branch (:160-161): up_layer = UpSampling2D(), no weights, the
    up-sampled tensor keeps its 2c channels, so the first decoder conv of a level sees 3c inputs.
    """
    L = []
    enc = [nfb, nfb * 2, nfb * 4, nfb * 8, nfb * 16]
    cin = 1
    for lvl, c in enumerate(enc):                      # :172-196
        tag = 'b' if lvl == 4 else 'e%d' % lvl
        L.append((tag + 'a', 'conv', cin, c, 0.99))
        L.append((tag + 'b', 'conv', c, c, 0.99))
        cin = c
    for lvl in (3, 2, 1, 0):                           # :197-220
        c = enc[lvl]
        if not upsampling:
            L.append(('u%d' % lvl, 'convT', c * 2, c, 0.5))  # up_layer :154-161
        L.append(('d%da' % lvl, 'conv', c * 3 if upsampling else c * 2, c, 0.99))     # after concat :200
        L.append(('d%db' % lvl, 'conv', c, c, 0.99))
    L.append(('out', 'head', nfb, 2, None))            # :221
    return L


def dropout_rates(drp=0.25):
    """Dropout rate after each named layer.  unet_2d_summary.py:179,185,191,198,204,210,216."""
    return {'e1b': drp, 'e2b': 2 * drp, 'e3b': 2 * drp,
            'u3': 2 * drp, 'u2': 2 * drp, 'u1': 2 * drp, 'u0': drp}


def weight_shapes(nfb=32, upsampling=False):
    """Shapes of the 134 (110 with upsampling) get_weights() arrays."""
    shapes = []
    for name, kind, cin, cout, _ in layer_table(nfb, upsampling):
        if kind == 'conv':
            shapes += [(3, 3, cin, cout), (cout,)] + [(cout,)] * 4
        elif kind == 'convT':
            shapes += [(2, 2, cout, cin), (cout,)] + [(cout,)] * 4
        else:
            shapes += [(1, 1, cin, cout), (cout,)]
    return shapes


def _trunc_normal(rs, shape, std):
    """TF truncated_normal: resample beyond 2 sigma (Appendix A.6)."""
    out = rs.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rs.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def init_weights(nfb=32, seed=7535, dtype=np.float32, randomize_bn=False, upsampling=False):
    """he_normal conv kernels (fan_in = prod(shape[:-2])*shape[-2]), glorot-uniform head, BN identity."""
    rs = np.random.RandomState(seed)
    W = []
    for name, kind, cin, cout, _ in layer_table(nfb, upsampling):
        if kind == 'conv':
            W.append(_trunc_normal(rs, (3, 3, cin, cout), np.sqrt(2.0 / (9 * cin))))
        elif kind == 'convT':
            # Keras fan_in for a (2,2,Cout,Cin) kernel is 4*Cout (quirk, Appendix A.5).
            W.append(_trunc_normal(rs, (2, 2, cout, cin), np.sqrt(2.0 / (4 * cout))))
        else:
            lim = np.sqrt(6.0 / (cin + cout))
            W.append(rs.uniform(-lim, lim, (1, 1, cin, cout)))
        W.append(np.zeros(cout))
        if kind != 'head':
            if randomize_bn:
                W += [rs.uniform(0.5, 1.5, cout), rs.uniform(-0.3, 0.3, cout),
                      rs.uniform(-0.2, 0.2, cout), rs.uniform(0.5, 1.5, cout)]
                W[-5] = rs.uniform(-0.1, 0.1, cout)  # non-zero conv bias
            else:
                W += [np.ones(cout), np.zeros(cout), np.zeros(cout), np.ones(cout)]
    return [w.astype(dtype) for w in W]


# ----------------------------------------------------------------------------
# Ops
# ----------------------------------------------------------------------------
def conv3x3_fwd(x, K, b):
    """'same' 3x3 stride-1 cross-correlation, zero pad 1 (A.2)."""
    N, H, W, Ci = x.shape
    xp = np.zeros((N, H + 2, W + 2, Ci), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    z = np.zeros((N, H, W, K.shape[3]), x.dtype)
    for a in range(3):
        for c in range(3):
            z += xp[:, a:a + H, c:c + W, :] @ K[a, c]
    return z + b


def conv3x3_bwd(x, K, dz):
    """Returns dx, dK, db for conv3x3_fwd."""
    N, H, W, Ci = x.shape
    Co = K.shape[3]
    xp = np.zeros((N, H + 2, W + 2, Ci), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    dxp = np.zeros_like(xp)
    dK = np.zeros_like(K)
    dz2 = dz.reshape(-1, Co)
    for a in range(3):
        for c in range(3):
            win = xp[:, a:a + H, c:c + W, :]
            dK[a, c] = win.reshape(-1, Ci).T @ dz2
            dxp[:, a:a + H, c:c + W, :] += dz @ K[a, c].T
    return dxp[:, 1:-1, 1:-1], dK, dz2.sum(0)


def convT2x2_fwd(x, K, b):
    """Conv2DTranspose(k=2, s=2, 'valid'); K is (2,2,Cout,Cin) (A.5)."""
    N, H, W, Ci = x.shape
    Co = K.shape[2]
    out = np.zeros((N, 2 * H, 2 * W, Co), x.dtype)
    for a in range(2):
        for c in range(2):
            out[:, a::2, c::2, :] = x @ K[a, c].T
    return out + b


def convT2x2_bwd(x, K, dz):
    dx = np.zeros_like(x)
    dK = np.zeros_like(K)
    Ci = x.shape[3]
    Co = K.shape[2]
    for a in range(2):
        for c in range(2):
            d = dz[:, a::2, c::2, :]
            dx += d @ K[a, c]
            dK[a, c] = d.reshape(-1, Co).T @ x.reshape(-1, Ci)
    return dx, dK, dz.reshape(-1, Co).sum(0)


def bn_train_fwd(z, gamma, beta, eps=BN_EPS):
    """Batch statistics over (N,H,W); population (biased) variance (A.3)."""
    mu = z.mean(axis=(0, 1, 2))
    var = ((z - mu) ** 2).mean(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (z - mu) * inv
    return xhat * gamma + beta, (xhat, inv, mu, var)


def bn_train_bwd(dy, gamma, cache):
    xhat, inv = cache[0], cache[1]
    M = dy.shape[0] * dy.shape[1] * dy.shape[2]
    dbeta = dy.sum(axis=(0, 1, 2))
    dgamma = (dy * xhat).sum(axis=(0, 1, 2))
    dz = gamma * inv * (dy - dbeta / M - xhat * dgamma / M)
    return dz, dgamma, dbeta


def bn_infer(z, gamma, beta, mmean, mvar, eps=BN_EPS):
    return (z - mmean) / np.sqrt(mvar + eps) * gamma + beta


def maxpool2x2_fwd(x, forced_idx=None):
    """2x2/2 'valid' max pool; argmax = FIRST max in row-major window order (A.7).
    forced_idx (N,H/2,W/2,C) in 0..3: route through THESE window positions instead of the float64 argmax (see
    UNetOracle `force`): a window whose top two values differ by less than fp32 rounding has no well-defined winner."""
    N, H, W, C = x.shape
    win = np.stack([x[:, 0::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 0::2], x[:, 1::2, 1::2]], axis=-1)
    if forced_idx is not None:
        idx = np.asarray(forced_idx, np.uint8)
        return np.take_along_axis(win, idx[..., None].astype(np.intp), axis=-1)[..., 0], idx
    idx = win.argmax(axis=-1).astype(np.uint8)        # numpy argmax returns the first max
    return win.max(axis=-1), idx


def maxpool2x2_bwd(dy, idx):
    N, h, w, C = dy.shape
    dx = np.zeros((N, 2 * h, 2 * w, C), dy.dtype)
    for k, (a, c) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        dx[:, a::2, c::2, :] = dy * (idx == k)
    return dx


def head_fwd(a, K, b):
    """1x1 conv to 2 logits + max-subtracted softmax, keep class 1 (:221-222, A.9)."""
    logits = a @ K[0, 0] + b
    e = np.exp(logits - logits.max(axis=-1, keepdims=True))
    sm = e / e.sum(axis=-1, keepdims=True)
    return sm[..., 1], sm


def bce_keras(p, y):
    """keras.losses.binary_crossentropy with the TF backend (A.10): clip, logit, stable BCE, global mean."""
    pc = np.clip(p, CLIP_LO, CLIP_HI)
    x = np.log(pc / (1 - pc))
    l = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    return l.mean()


def bce_keras_grad(p, y):
    """d(mean BCE)/dp; zero where the clip is active (A.10)."""
    inside = (p > CLIP_LO) & (p < CLIP_HI)
    pc = np.clip(p, CLIP_LO, CLIP_HI)
    # l(x) with x = logit(pc): dl/dx = sigmoid(x) - y = pc - y ; dx/dp = 1/(pc(1-pc))
    return inside * (pc - y) / (pc * (1 - pc)) / p.size


def alt_loss(name, p, y):
    """The reference's selectable alternates (unet_2d_summary.py:372-377 -> utils/neurons.py:13-29,78-94) as
    (scalar loss, dL/dp).  Keras means the weighted-BCE matrix over everything; dice losses are already scalars."""
    if name == 'weighted_binary_crossentropy':
        l = -(2.0 * y * np.log(p + 1e-7) + (1 - y) * np.log(1 - p + 1e-7))
        return l.mean(), -(2.0 * y / (p + 1e-7) - (1 - y) / (1 - p + 1e-7)) / p.size
    inter = (y * p).sum()
    if name == 'dice_loss':
        D = y.sum() + p.sum() + 1e-7
        return 1 - 2 * inter / D, -2 * (y * D - inter) / D ** 2
    if name == 'dicesq_loss':
        D = (y ** 2).sum() + (p ** 2).sum() + K_EPS
        return -2 * inter / D, -2 * (y * D - 2 * p * inter) / D ** 2
    raise ValueError(name)


def keras_metrics(y, p):
    """The 7 compile() metrics, /root/reference/deepcalcium/utils/neurons.py:32-50,70-75,86-90,97-106."""
    y = y.astype(np.float64)
    pr = np.round(p)                                   # numpy rounds half to even, like K.round
    tp = (pr * y).sum()
    prec = tp / (pr.sum() + K_EPS)
    fn = np.clip(y - pr, 0, 1).sum()
    reca = tp / (tp + fn + K_EPS)
    f1 = 2 * prec * reca / (prec + reca + K_EPS)
    dice = 2 * tp / (y.sum() + pr.sum() + 1e-7)
    dicesq = 2 * (y * p).sum() / ((y ** 2).sum() + (p ** 2).sum() + K_EPS)
    posyt = y.sum() / (y.size + K_EPS)
    posyp = pr.sum() / (p.size + K_EPS)
    return dict(F1=f1, prec=prec, reca=reca, dice=dice, dicesq=dicesq, posyt=posyt, posyp=posyp)


def adam_keras(p, g, m, v, it, lr=0.002, b1=0.9, b2=0.999, eps=1e-8):
    """Keras-2.0.6 Adam (A.11 / SURVEY a10): eps OUTSIDE the bias correction. `it` = iterations before this step."""
    t = it + 1
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m2 = b1 * m + (1 - b1) * g
    v2 = b2 * v + (1 - b2) * g * g
    return p - lr_t * m2 / (np.sqrt(v2) + eps), m2, v2


# ----------------------------------------------------------------------------
# Whole network
# ----------------------------------------------------------------------------
class UNetOracle(object):
    """Float64 forward/backward of unet() (unet_2d_summary.py:123-224) on a Keras-ordered weight list.

    force = {'gates': {layer: bool (N,h,w,C)}, 'pool': {lvl: uint8 (N,h/2,w/2,C)}} (either key optional) pins the two
    DISCONTINUOUS decisions of the network -- which side of 0 a pre-activation falls (Activation('relu'),
    unet_2d_summary.py:167,:159) and which element of a 2x2 window is the maximum (MaxPooling2D, :176) -- to the ones
    another implementation took, the way explicit dropout masks pin Dropout.  With them pinned the function is smooth
    in its arithmetic, so an fp32 implementation and this float64 one differ by rounding only and gradients can be held
    to ~1e-4 instead of being compared in a loose norm (a pre-activation within fp32 rounding of 0 flips one gate per
    10^5-10^6 elements and moves the affected gradient entries by O(1 %)).  A forced gate multiplies: a = y * gate."""

    def __init__(self, weights, nfb=32, drp=0.25, dtype=np.float64, upsampling=False, force=None):
        self.force = force or {}
        self.nfb = nfb
        self.upsampling = upsampling
        self.table = layer_table(nfb, upsampling)
        self.drop = dropout_rates(drp) if drp else {}
        self.dtype = dtype
        self.P = {}
        i = 0
        for name, kind, cin, cout, mom in self.table:
            n = 2 if kind == 'head' else 6
            self.P[name] = [np.asarray(w, dtype) for w in weights[i:i + n]]
            i += n
        assert i == len(weights)

    def weights(self):
        out = []
        for name, *_ in self.table:
            out += self.P[name]
        return out

    # -- one conv/convT + BN + ReLU (+dropout) block --------------------------
    def _block(self, name, kind, x, training, masks, cache):
        p = self.P[name]
        z = conv3x3_fwd(x, p[0], p[1]) if kind == 'conv' else convT2x2_fwd(x, p[0], p[1])
        if training:
            y, bnc = bn_train_fwd(z, p[2], p[3])
        else:
            y, bnc = bn_infer(z, p[2], p[3], p[4], p[5]), None
        fg = self.force.get('gates') if training else None
        gate = np.asarray(fg[name], bool) if fg is not None else y > 0
        a = np.where(gate, y, 0.0) if fg is not None else np.maximum(y, 0)
        keep = None
        if training and name in self.drop and self.drop[name] > 0:
            keep = 1.0 - self.drop[name]
            a = a * masks[name].astype(self.dtype) / keep      # A.8
        if cache is not None:
            cache[name] = (x, bnc, gate, keep)
        return a

    def forward(self, x, training=False, masks=None, cache=None, taps=None):
        x = np.asarray(x, self.dtype)[..., None]           # Lambda expand_dims :170
        skips = {}
        t = {n: k for n, k, *_ in self.table}
        for lvl in range(5):
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            x = self._block(tag + 'a', 'conv', x, training, masks, cache)
            x = self._block(tag + 'b', 'conv', x, training, masks, cache)
            if taps is not None:
                taps[tag + 'b'] = x
            if lvl < 4:
                skips[lvl] = x                             # dropped-out tensor is the skip (:179-180)
                fp = self.force.get('pool') if training else None
                x, idx = maxpool2x2_fwd(x, fp[lvl] if fp is not None else None)
                if cache is not None:
                    cache['p%d' % lvl] = idx
                if taps is not None:
                    taps['p%d' % lvl] = x
                    taps['p%d_idx' % lvl] = idx
        for lvl in (3, 2, 1, 0):
            if self.upsampling:                            # UpSampling2D() then Dropout (:160-161, :198)
                x = np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)
                name = 'u%d' % lvl
                if training and self.drop.get(name, 0) > 0:
                    x = x * masks[name].astype(self.dtype) / (1.0 - self.drop[name])
            else:
                x = self._block('u%d' % lvl, 'convT', x, training, masks, cache)
            x = np.concatenate([x, skips[lvl]], axis=-1)   # up path first (:200)
            x = self._block('d%da' % lvl, 'conv', x, training, masks, cache)
            x = self._block('d%db' % lvl, 'conv', x, training, masks, cache)
            if taps is not None:
                taps['d%db' % lvl] = x
        p, sm = head_fwd(x, *self.P['out'])
        if cache is not None:
            cache['out'] = (x, sm)
        return p

    def loss_and_grads(self, x, y, masks=None, taps=None, loss='binary_crossentropy'):
        """One training-mode forward + backward.  Returns (loss, p, grads{name:[dK,db,dgamma,dbeta]}, batch_stats)."""
        cache = {}
        p = self.forward(x, True, masks, cache)
        yf = np.asarray(y, self.dtype)
        if loss == 'binary_crossentropy':
            loss, dp = bce_keras(p, yf), bce_keras_grad(p, yf)
        else:
            loss, dp = alt_loss(loss, p, yf)
        a, sm = cache['out']
        # p = sm1 ; dsm1/dz1 = sm1*sm0 ; dsm1/dz0 = -sm1*sm0
        s = dp * sm[..., 1] * sm[..., 0]
        dlog = np.stack([-s, s], axis=-1)
        Kh, bh = self.P['out']
        G = {'out': [(a.reshape(-1, a.shape[-1]).T @ dlog.reshape(-1, 2))[None, None], dlog.reshape(-1, 2).sum(0)]}
        da = dlog @ Kh[0, 0].T
        stats = {}

        def block_bwd(name, kind, da):
            xin, bnc, relu_mask, keep = cache[name]
            if keep is not None:
                da = da * cache['_masks'][name] / keep
            dy = da * relu_mask
            pl = self.P[name]
            dz, dg, db_ = bn_train_bwd(dy, pl[2], bnc)
            if taps is not None:
                taps['da_' + name], taps['dz_' + name], taps['x_' + name] = da, dz, xin
            if kind == 'conv':
                dx, dK, dbias = conv3x3_bwd(xin, pl[0], dz)
            else:
                dx, dK, dbias = convT2x2_bwd(xin, pl[0], dz)
            G[name] = [dK, dbias, dg, db_]
            stats[name] = (bnc[2], bnc[3])
            return dx

        cache['_masks'] = {k: np.asarray(v, self.dtype) for k, v in (masks or {}).items()}
        dskip = {}
        for lvl in (0, 1, 2, 3):
            da = block_bwd('d%db' % lvl, 'conv', da)
            dcat = block_bwd('d%da' % lvl, 'conv', da)
            if self.upsampling:
                C = dcat.shape[-1] // 3
                dskip[lvl] = dcat[..., 2 * C:]
                name = 'u%d' % lvl
                dup = dcat[..., :2 * C]
                if self.drop.get(name, 0) > 0:
                    dup = dup * cache['_masks'][name] / (1.0 - self.drop[name])
                da = dup[:, 0::2, 0::2] + dup[:, 0::2, 1::2] + dup[:, 1::2, 0::2] + dup[:, 1::2, 1::2]
                continue
            C = dcat.shape[-1] // 2
            dskip[lvl] = dcat[..., C:]
            da = block_bwd('u%d' % lvl, 'convT', dcat[..., :C])
        for lvl in (4, 3, 2, 1, 0):
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            if lvl < 4:
                da = maxpool2x2_bwd(da, cache['p%d' % lvl]) + dskip[lvl]
            da = block_bwd(tag + 'b', 'conv', da)
            da = block_bwd(tag + 'a', 'conv', da)
        return loss, p, G, stats

    def train_step(self, x, y, opt_state, masks=None, lr=0.002):
        """fwd + bwd + Keras Adam + BN moving-average update (A.3, A.11).  opt_state = dict(it, m{}, v{})."""
        loss, p, G, stats = self.loss_and_grads(x, y, masks)
        it = opt_state['it']
        for name, kind, cin, cout, mom in self.table:
            pl = self.P[name]
            for j, g in enumerate(G[name]):
                key = (name, j)
                m = opt_state['m'].get(key, np.zeros_like(pl[j]))
                v = opt_state['v'].get(key, np.zeros_like(pl[j]))
                pl[j], opt_state['m'][key], opt_state['v'][key] = adam_keras(pl[j], g.reshape(pl[j].shape), m, v, it, lr)
            if kind != 'head':
                mu, var = stats[name]
                pl[4] = pl[4] * mom + mu * (1 - mom)
                pl[5] = pl[5] * mom + var * (1 - mom)       # biased variance, no Bessel (A.3)
        opt_state['it'] = it + 1
        return loss, p


def make_drop_masks(nfb, N, H, W, drp=0.25, seed=7, upsampling=False):
    """Explicit Bernoulli(keep) masks, identical bits for oracle and HIP path."""
    rs = np.random.RandomState(seed)
    out = {}
    k = 2 if upsampling else 1          # the up-sampled tensor keeps the 2c channels of the level below
    shp = {'e1b': (H // 2, nfb * 2), 'e2b': (H // 4, nfb * 4), 'e3b': (H // 8, nfb * 8),
           'u3': (H // 8, nfb * 8 * k), 'u2': (H // 4, nfb * 4 * k), 'u1': (H // 2, nfb * 2 * k), 'u0': (H, nfb * k)}
    for name, rate in dropout_rates(drp).items():
        h, c = shp[name]
        w = h * W // H
        out[name] = (rs.random_sample((N, h, w, c)) < (1.0 - rate)).astype(np.uint8)
    return out


def synthetic_batch(N, H, W, seed_x=865, seed_y=866, pos_rate=0.126):
    """SURVEY 8(d) synthetic inputs: x ~ N(0,1), y ~ Bernoulli(0.126)."""
    x = np.random.RandomState(seed_x).standard_normal((N, H, W)).astype(np.float32)
    y = (np.random.RandomState(seed_y).random_sample((N, H, W)) < pos_rate).astype(np.uint8)
    return x, y
