"""Test helper: the two DISCONTINUOUS decisions the engine's last forward_train took -- ReLU gates and max-pool argmax
indices -- read back so that the float64 oracle can be run through the same ones (`force=` of oracle.unet_numpy.UNetOracle
/ oracle.unet_torch.UNetTorch), the way explicit dropout masks already pin Dropout.  What is left between the fp32 HIP
path and the float64 oracle is then pure rounding and the gradients can be held to ~1e-4 per tensor.

Gate of layer l, element e: the sign of the exact value of fmaf(z[e], sc, sh) with (sc, sh) = dc_bn_affine(mean, invstd,
gamma, beta) (csrc/common.h) -- the one expression every kernel that applies or differentiates BatchNorm + ReLU
evaluates.  z (pre-BN) is kept for every layer; the product of two fp32 is exact in float64 and the addition cannot
change the sign, so the float64 evaluation below gives the device's bit."""
import numpy as np
import torch


def device_decisions(eng, N, weights=None):
    """-> dict(gates={layer: bool (N,h,w,C) numpy}, pool={lvl: uint8 (N,h/2,w/2,C) numpy}) of the last forward_train.
    weights: the Keras-ordered weight list that forward ran with, when the engine's parameters have moved since (after an
    Adam step gamma / beta in pflat are the NEW ones)."""
    T = eng._train_bufs(N)
    A = eng._acts(N)
    gates = {}
    for i, l in enumerate(eng.layers):
        if l.kind == 'head':
            continue
        o, c = eng._stat_off[l.name], l.cout
        mean, invstd = eng.bstat[o:o + c], eng.bstat[o + c:o + 2 * c]
        g0, b0 = l.off['gamma'][0], l.off['beta'][0]
        gamma, beta = eng.pflat[g0:g0 + c], eng.pflat[b0:b0 + c]
        if weights is not None:
            gamma = torch.as_tensor(np.asarray(weights[6 * i + 2], np.float32)).to(eng.device)
            beta = torch.as_tensor(np.asarray(weights[6 * i + 3], np.float32)).to(eng.device)
        sc = gamma * invstd                                                  # fp32 product, as the device forms it
        sh = (beta.double() - mean.double() * sc.double()).float()          # fmaf(-mean, sc, beta): one rounding
        z = T['z_' + l.name]
        h, w = eng._hw(l.lvl)
        zz = z.view(-1)[:N * h * w * c].view(N, h, w, c)
        gates[l.name] = ((zz.double() * sc.double() + sh.double()) > 0).cpu().numpy()
    pool = {lvl: A['idx%d' % lvl].cpu().numpy().copy() for lvl in range(4)}
    return dict(gates=gates, pool=pool)


def assert_forcing_is_benign(Wt, nfb, x, masks, p_forced, p_dev=None, tol=1e-5, **oracle_kw):
    """A forced-only comparison would copy a WRONG device decision (a bad training-path pool index, say) into the reference
    and pass.  Run the oracle UN-forced too: forcing the device's decisions may move the probabilities only by what a
    handful of fp32-vs-float64 near-ties can (< tol), and the device must also sit within 1e-4 of the un-forced ones."""
    from oracle import unet_numpy as on
    p_u = on.UNetOracle(Wt, nfb, **oracle_kw).forward(x, training=True, masks=masks)
    d = float(np.abs(np.asarray(p_forced) - p_u).max())
    assert d < tol, 'forcing the device decisions moved the oracle probabilities by %.3e' % d
    if p_dev is not None:
        assert float(np.abs(np.asarray(p_dev) - p_u).max()) < 1e-4
    return d


def flat_grads(G, ref):
    """Concatenate per-layer gradient lists in `ref`'s order, skipping the conv biases in front of BatchNorm (their
    gradient is analytically zero: both sides hold rounding noise there)."""
    return np.concatenate([np.asarray(g, np.float64).ravel() for k in ref for j, g in enumerate(G[k]) if not (j == 1 and k != 'out')])


def grad_report(G, G_ref, tag=''):
    """Per-tensor max|g - r| / max|r| and whole-gradient (cos, rel-L2); prints a one-line summary + the worst tensor."""
    worst, worst_name = 0.0, None
    for k in G_ref:
        for j, (g, r) in enumerate(zip(G[k], G_ref[k])):
            if j == 1 and k != 'out':
                continue
            r = np.asarray(r, np.float64).reshape(np.asarray(g).shape)
            e = np.abs(np.asarray(g, np.float64) - r).max() / max(np.abs(r).max(), 1e-30)
            if e > worst:
                worst, worst_name = e, '%s[%d]' % (k, j)
    fg, fr = flat_grads(G, G_ref), flat_grads(G_ref, G_ref)
    cos = float(fg.dot(fr) / (np.linalg.norm(fg) * np.linalg.norm(fr)))
    rel = float(np.linalg.norm(fg - fr) / np.linalg.norm(fr))
    print('%sgradient vs float64 oracle: worst per-tensor max|g-r|/max|r| = %.3e (%s), rel-L2 = %.3e, 1-cos = %.3e'
          % (tag, worst, worst_name, rel, 1.0 - cos))
    return worst, rel, cos


def count_flips(dec, cache, masks):
    """ReLU gates that differ between the device and an UNFORCED oracle run (its cache), dropped elements excluded."""
    flips = 0
    for name, gate in dec['gates'].items():
        ref = cache[name][2]
        live = masks[name].astype(bool) if masks is not None and name in masks else np.ones_like(ref)
        flips += int(((gate != ref) & live).sum())
    return flips
