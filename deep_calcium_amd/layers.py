"""The weighted layers of the UNet2DS graph and the flat parameter layout (torch-free: the engine, the checkpoint reader and
the background checkpoint-writer process all use it).  Graph order = Keras get_weights() order,
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:172-221."""
import numpy as np


class LayerSpec(object):
    __slots__ = ('name', 'kind', 'cin', 'cout', 'mom', 'lvl', 'drop', 'off', 'soff', 'index')

    def __init__(self, name, kind, cin, cout, mom, lvl, drop, index):
        self.name, self.kind, self.cin, self.cout, self.mom, self.lvl, self.drop, self.index = \
            name, kind, cin, cout, mom, lvl, drop, index
        self.off = {}    # trainable: 'k','b','gamma','beta' -> (offset, shape) in pflat
        self.soff = {}   # moving stats: 'mmean','mvar' -> offset in sflat

    @property
    def kshape(self):
        if self.kind == 'conv':
            return (3, 3, self.cin, self.cout)
        if self.kind == 'convT':
            return (2, 2, self.cout, self.cin)
        return (1, 1, self.cin, self.cout)


def build_layer_table(nfb=32, drp=0.25, upsampling=False):
    """Weighted layers in graph-creation order (= Keras get_weights order), unet_2d_summary.py:172-221.
    upsampling=True: the UpSampling2D branch (:160-161) -- no up-conv layers, first decoder conv sees 3c inputs."""
    enc = [nfb << i for i in range(5)]
    rates = {'e1b': drp, 'e2b': 2 * drp, 'e3b': 2 * drp, 'u3': 2 * drp, 'u2': 2 * drp, 'u1': 2 * drp, 'u0': drp}
    L = []
    cin = 1
    for lvl, c in enumerate(enc):
        tag = 'b' if lvl == 4 else 'e%d' % lvl
        for sfx, ci in (('a', cin), ('b', c)):
            L.append(LayerSpec(tag + sfx, 'conv', ci, c, 0.99, lvl, rates.get(tag + sfx, 0.0), len(L)))
        cin = c
    for lvl in (3, 2, 1, 0):
        c = enc[lvl]
        if not upsampling:
            L.append(LayerSpec('u%d' % lvl, 'convT', 2 * c, c, 0.5, lvl, rates.get('u%d' % lvl, 0.0), len(L)))
        L.append(LayerSpec('d%da' % lvl, 'conv', 3 * c if upsampling else 2 * c, c, 0.99, lvl, 0.0, len(L)))
        L.append(LayerSpec('d%db' % lvl, 'conv', c, c, 0.99, lvl, 0.0, len(L)))
    L.append(LayerSpec('out', 'head', nfb, 2, None, 0, 0.0, len(L)))
    return L


def assign_offsets(layers):
    """Sets l.off ('k','b','gamma','beta' -> (offset, shape) in the flat parameter buffer: kernel, bias, gamma, beta per layer
    in graph order) and l.soff ('mmean','mvar' -> offset in the flat moving-statistics buffer).  -> (n_train, n_stats)."""
    off = soff = 0
    for l in layers:
        l.off['k'] = (off, l.kshape); off += int(np.prod(l.kshape))
        l.off['b'] = (off, (l.cout,)); off += l.cout
        if l.kind != 'head':
            l.off['gamma'] = (off, (l.cout,)); off += l.cout
            l.off['beta'] = (off, (l.cout,)); off += l.cout
            l.soff['mmean'] = soff; soff += l.cout
            l.soff['mvar'] = soff; soff += l.cout
    return off, soff


def split_weights(layers, p, s):
    """Flat parameter / moving-statistics arrays -> Keras get_weights(): [kernel, bias, gamma, beta, moving_mean,
    moving_variance] per layer (134 arrays for the transpose network)."""
    out = []
    for l in layers:
        for key in ('k', 'b', 'gamma', 'beta'):
            if key in l.off:
                o, shp = l.off[key]
                out.append(p[o:o + int(np.prod(shp))].reshape(shp).copy())
        if l.kind != 'head':
            out.append(s[l.soff['mmean']:l.soff['mmean'] + l.cout].copy())
            out.append(s[l.soff['mvar']:l.soff['mvar'] + l.cout].copy())
    return out


def split_optimizer(layers, m, v):
    """Flat Adam moments -> per-weight lists in trainable_weights order (what keras_io writes as optimizer_weights)."""
    ms, vs = [], []
    for l in layers:
        for key in ('k', 'b', 'gamma', 'beta'):
            if key in l.off:
                o, shp = l.off[key]
                n = int(np.prod(shp))
                ms.append(m[o:o + n].reshape(shp).copy())
                vs.append(v[o:o + n].reshape(shp).copy())
    return ms, vs
