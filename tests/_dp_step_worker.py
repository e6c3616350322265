"""Worker of tests/test_dp_gpu.py (one process per rank): one data-parallel optimizer step through the product API
(`Model.train_on_batch` on the GLOBAL batch: shard -> forward -> backward with the bucketed gradient all-reduce -> Adam),
'local' BatchNorm, checked against the oracle run as independent shards with averaged gradients."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import parallel                      # noqa: E402
from deep_calcium_amd.model import Model, Adam             # noqa: E402
from oracle import unet_numpy as on                        # noqa: E402
from _forced import device_decisions, grad_report          # noqa: E402


def flat(G, ref):
    return np.concatenate([g.ravel() for n in ref for j, g in enumerate(G[n]) if not (j == 1 and n != 'out')]).astype(np.float64)


def main():
    out_path = sys.argv[1]
    rank, world = parallel.init_from_env()
    NG, H, W, nfb = 2 * world, 32, 32, 8
    Wt = on.init_weights(nfb, seed=99, randomize_bn=True)
    x, y = on.synthetic_batch(NG, H, W)
    masks = on.make_drop_masks(nfb, NG, H, W)
    res = {}
    for buckets in ('3', '1'):
        os.environ['DC_AR_BUCKETS'] = buckets
        model = Model((H, W), nfb)
        model.compile(Adam(0.002), 'binary_crossentropy')
        model.set_weights(Wt)
        vals = model.train_on_batch(x, y, drop_masks=masks)
        torch.cuda.synchronize()
        eng = model.engine
        res[buckets] = (vals, eng.gflat.cpu().numpy().copy(), eng.pflat.cpu().numpy().copy(),
                        {k: [g / world for g in v] for k, v in eng.grads().items()})
        # the ReLU gates / pool indices THIS rank's shard step took (gamma / beta as they were before the Adam step)
        dec = device_decisions(eng, NG // world, weights=Wt)
    decs = [None] * world
    torch.distributed.gather_object(dec, decs if rank == 0 else None, dst=0)
    # the RNG-dropout path: every rank must draw the bits ONE device would have drawn for its slice of the global batch
    seeds = [parallel.shard_drop_seed(12345, 1000, r) for r in range(world)]
    # numpy RNG broadcast: ranks start from different states, end on rank 0's
    np.random.seed(100 + rank)
    parallel.broadcast_numpy_rng()
    draw = float(np.random.random_sample())
    t = torch.tensor([draw], dtype=torch.float64, device='cuda')
    lo, hi = t.clone(), t.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        gs, ls = [], []
        ref = None
        for r in range(world):
            s_r = parallel.shard_slice(NG, r, world)
            # the oracle's shard step through rank r's own gates / pool routes: what is compared is rounding only
            l_r, p_r, G_r, _ = on.UNetOracle(Wt, nfb, force=decs[r]).loss_and_grads(x[s_r], y[s_r], {k: v[s_r] for k, v in masks.items()})
            ref = G_r
            gs.append(G_r)
            ls.append(l_r)
        G_mean = {k: [np.mean([g[k][j] for g in gs], 0) for j in range(len(gs[0][k]))] for k in ref}
        vals, g3, p3, G3 = res['3']
        _, g1, p1, _ = res['1']
        worst, rel, cos = grad_report(G3, G_mean, 'DP local mode, %d ranks, forced gates: ' % world)
        out = dict(world=world, loss_err=abs(vals[0] - float(np.mean(ls))),
                   grad_rel=rel, grad_worst=worst,
                   buckets_bitwise_grad=bool(np.array_equal(g3, g1)), buckets_bitwise_params=bool(np.array_equal(p3, p1)),
                   seeds_distinct=len(set(seeds)) == world, rng_same=bool(lo.item() == hi.item()))
        json.dump(out, open(out_path, 'w'))
    parallel.barrier()


if __name__ == '__main__':
    main()
