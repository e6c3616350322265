"""Oracle B: torch-CPU functional restatement of the UNet2DS arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); PARITY UNPINNED by the
reference (Keras/TF un-vendored).  Two uses:
  * float64 + autograd: an independent second opinion on oracle A's
    hand-derived backward passes (tests/test_oracle.py, agreement 1e-10);
  * float32 + all host threads (oneDNN): the CPU baseline `bench.py` times
    beside the GPU ("cpu_baseline.kind" = "port"), standing in for the
    reference's Keras-CPU path which cannot be installed offline
    (BASELINE.md section 3).
Topology follows /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:123-224.
BatchNorm is written out by hand because torch's running variance is unbiased
while Keras 2.0.6's is biased.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .unet_numpy import layer_table, dropout_rates, BN_EPS, CLIP_LO, CLIP_HI


def _to_t(w, dtype, requires_grad=False):
    t = torch.tensor(np.asarray(w), dtype=dtype)
    t.requires_grad_(requires_grad)
    return t


class UNetTorch(object):
    def __init__(self, weights, nfb=32, drp=0.25, dtype=torch.float64, requires_grad=True, upsampling=False, force=None):
        # force: see oracle.unet_numpy.UNetOracle -- ReLU gates {layer: bool NHWC} and pool indices {lvl: uint8 NHWC}
        # taken from another implementation, applied in training mode (autograd then routes through exactly them)
        self.force = force or {}
        self.upsampling = upsampling
        self.table = layer_table(nfb, upsampling)
        self.drop = dropout_rates(drp) if drp else {}
        self.dtype = dtype
        self.P = {}
        i = 0
        for name, kind, cin, cout, mom in self.table:
            n = 2 if kind == 'head' else 6
            self.P[name] = [_to_t(w, dtype, requires_grad and j < 4) for j, w in enumerate(weights[i:i + n])]
            i += n

    def _bn_relu(self, name, z, training, stats):
        g, b, mm, mv = self.P[name][2:6]
        if training:
            mu = z.mean(dim=(0, 2, 3), keepdim=True)
            var = ((z - mu) ** 2).mean(dim=(0, 2, 3), keepdim=True)     # biased
            if stats is not None:
                stats[name] = (mu.flatten().detach(), var.flatten().detach())
        else:
            mu, var = mm.view(1, -1, 1, 1), mv.view(1, -1, 1, 1)
        y = (z - mu) / torch.sqrt(var + BN_EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        fg = self.force.get('gates') if training else None
        if fg is not None:
            gate = torch.as_tensor(np.asarray(fg[name])).to(torch.bool).permute(0, 3, 1, 2)
            return torch.where(gate, y, torch.zeros((), dtype=y.dtype))
        return F.relu(y)

    def _block(self, name, kind, x, training, masks, stats):
        K, b = self.P[name][0], self.P[name][1]
        if kind == 'conv':
            z = F.conv2d(x, K.permute(3, 2, 0, 1), b, padding=1)           # HWIO -> OIHW
        else:
            z = F.conv_transpose2d(x, K.permute(3, 2, 0, 1), b, stride=2)   # (2,2,Co,Ci) -> (Ci,Co,2,2)
        a = self._bn_relu(name, z, training, stats)
        if training and name in self.drop and self.drop[name] > 0:
            keep = 1.0 - self.drop[name]
            m = torch.as_tensor(np.asarray(masks[name]), dtype=self.dtype).permute(0, 3, 1, 2)
            a = a * m / keep
        return a

    def forward(self, x, training=False, masks=None, stats=None, taps=None):
        """x: (N,H,W) -> p: (N,H,W).  Internals are NCHW (torch's native CPU layout).
        taps (dict): receives 'skip<lvl>' = the tensor each max-pool reads (detached, NCHW) for pool-index checks."""
        x = torch.as_tensor(np.asarray(x), dtype=self.dtype)[:, None]
        skips = {}
        for lvl in range(5):
            tag = 'b' if lvl == 4 else 'e%d' % lvl
            x = self._block(tag + 'a', 'conv', x, training, masks, stats)
            x = self._block(tag + 'b', 'conv', x, training, masks, stats)
            if lvl < 4:
                skips[lvl] = x
                if taps is not None:
                    taps['skip%d' % lvl] = x.detach()
                fp = self.force.get('pool') if training else None
                if fp is not None:
                    n_, c_, h_, w_ = x.shape
                    win = x.reshape(n_, c_, h_ // 2, 2, w_ // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n_, c_, h_ // 2, w_ // 2, 4)
                    idx = torch.as_tensor(np.asarray(fp[lvl])).to(torch.int64).permute(0, 3, 1, 2)
                    x = win.gather(-1, idx[..., None])[..., 0]
                else:
                    x = F.max_pool2d(x, 2, 2)
        for lvl in (3, 2, 1, 0):
            if self.upsampling:
                x = F.interpolate(x, scale_factor=2, mode='nearest')
                name = 'u%d' % lvl
                if training and self.drop.get(name, 0) > 0:
                    m = torch.as_tensor(np.asarray(masks[name]), dtype=self.dtype).permute(0, 3, 1, 2)
                    x = x * m / (1.0 - self.drop[name])
            else:
                x = self._block('u%d' % lvl, 'convT', x, training, masks, stats)
            x = torch.cat([x, skips[lvl]], dim=1)
            x = self._block('d%da' % lvl, 'conv', x, training, masks, stats)
            x = self._block('d%db' % lvl, 'conv', x, training, masks, stats)
        Kh, bh = self.P['out']
        logits = F.conv2d(x, Kh.permute(3, 2, 0, 1), bh)
        return torch.softmax(logits, dim=1)[:, 1]

    @staticmethod
    def bce(p, y):
        pc = torch.clamp(p, CLIP_LO, CLIP_HI)
        x = torch.log(pc / (1 - pc))
        # TF's sigmoid_cross_entropy_with_logits selects with where(x >= 0, ...) rather than relu/abs, so
        # its autograd is exact at x == 0 (p == 0.5 exactly, e.g. all-zero activations with a zero head
        # bias); relu/abs sub-gradients would be off by 0.5 there (caught by finite differences).
        pos = x >= 0
        zero = torch.zeros_like(x)
        return (torch.where(pos, x, zero) - x * y + torch.log1p(torch.exp(torch.where(pos, -x, x)))).mean()

    def loss_and_grads(self, x, y, masks=None, taps=None):
        stats = {}
        for pl in self.P.values():
            for t in pl:
                t.grad = None
        p = self.forward(x, True, masks, stats, taps)
        loss = self.bce(p, torch.as_tensor(np.asarray(y), dtype=self.dtype))
        loss.backward()
        G = {name: [t.grad.numpy() for t in pl[:4] if t.grad is not None] for name, pl in self.P.items()}
        return float(loss.detach()), p.detach().numpy(), G, {k: (a.numpy(), b.numpy()) for k, (a, b) in stats.items()}

    def train_step(self, x, y, opt_state, masks=None, lr=0.002, b1=0.9, b2=0.999, eps=1e-8):
        """fwd + bwd + Keras-form Adam + BN moving stats, in place (used as the timed CPU baseline)."""
        loss, p, _, stats = self.loss_and_grads(x, y, masks)
        it = opt_state['it']
        t = it + 1
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        with torch.no_grad():
            for name, kind, cin, cout, mom in self.table:
                pl = self.P[name]
                for j in range(4 if kind != 'head' else 2):
                    g = pl[j].grad
                    key = (name, j)
                    if key not in opt_state['m']:
                        opt_state['m'][key] = torch.zeros_like(pl[j])
                        opt_state['v'][key] = torch.zeros_like(pl[j])
                    m, v = opt_state['m'][key], opt_state['v'][key]
                    m.mul_(b1).add_(g, alpha=1 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1 - b2)
                    pl[j].sub_(lr_t * m / (v.sqrt() + eps))
                if kind != 'head':
                    mu, var = stats[name]
                    pl[4].mul_(mom).add_(torch.as_tensor(mu, dtype=self.dtype), alpha=1 - mom)
                    pl[5].mul_(mom).add_(torch.as_tensor(var, dtype=self.dtype), alpha=1 - mom)
        opt_state['it'] = it + 1
        return loss, p
