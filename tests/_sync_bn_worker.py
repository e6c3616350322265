"""Worker of tests/test_sync_bn_gpu.py (one process per rank, launched by torch.distributed.run): 'sync' BatchNorm over
G ranks x B images must reproduce ONE device's train step on the G*B batch -- checked against the float64 oracle."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import parallel                      # noqa: E402
from deep_calcium_amd.net import UNetEngine                # noqa: E402
from oracle import unet_numpy as on                        # noqa: E402
from _forced import device_decisions, grad_report          # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = parallel.init_from_env()
    NG, H, W, nfb = 2 * world, 32, 32, int(os.environ.get('DC_TEST_NFB', '8'))     # nfb 32: the role-split / dz-on-load / joint kernels engage
    Wt = on.init_weights(nfb, seed=99, randomize_bn=True)
    x, y = on.synthetic_batch(NG, H, W)
    masks = on.make_drop_masks(nfb, NG, H, W)
    sl = parallel.shard_slice(NG)
    res = {}
    for mode in ('sync', 'local'):
        eng = UNetEngine((H, W), nb_filters_base=nfb, bn_mode=mode)
        eng.set_weights(Wt)
        xd, yd = torch.from_numpy(x[sl]).cuda(), torch.from_numpy(y[sl]).cuda()
        md = {k: torch.from_numpy(np.ascontiguousarray(v[sl])).cuda() for k, v in masks.items()}
        p = eng.forward_train(xd, yd, md, update_moving=True).cpu().numpy()
        sums = eng._train_bufs(xd.shape[0])['sums'].clone()
        eng.backward()
        parallel.all_reduce_sum(eng.gflat)
        parallel.all_reduce_sum(sums)
        torch.cuda.synchronize()
        G = {k: [g / world for g in v] for k, v in eng.grads().items()}     # what Adam's grad_scale = 1/world applies
        decs = [None] * world             # every rank's ReLU gates / pool indices of this step, gathered on rank 0
        torch.distributed.gather_object(device_decisions(eng, xd.shape[0]), decs if rank == 0 else None, dst=0)
        res[mode] = (p, float(sums[0].item()) / (NG * H * W), G, eng.get_weights(), decs)
    if rank == 0:
        p, loss, G, Wn, decs = res['sync']
        # 'sync' = ONE device's step on the whole batch: the oracle's batch-4 step through the ranks' gates, concatenated
        whole = dict(gates={k: np.concatenate([d['gates'][k] for d in decs]) for k in decs[0]['gates']},
                     pool={k: np.concatenate([d['pool'][k] for d in decs]) for k in decs[0]['pool']})
        loss_ref, p_ref, G_ref, _ = on.UNetOracle(Wt, nfb, force=whole).loss_and_grads(x, y, masks)
        worst, rel, cos = grad_report(G, G_ref, "'sync' BatchNorm, %d ranks, forced gates: " % world)
        fr = np.concatenate([g.ravel() for n in G_ref for j, g in enumerate(G_ref[n]) if not (j == 1 and n != 'out')])
        pl, lossl, Gl, _, decl = res['local']
        fl = np.concatenate([g.ravel() for n in G_ref for j, g in enumerate(Gl[n]) if not (j == 1 and n != 'out')]).astype(np.float64)
        # 'local' mode = G independent shard steps (each shard its own BatchNorm statistics), gradients averaged
        gs, ls, ps = [], [], []
        for r in range(world):
            s_r = parallel.shard_slice(NG, r, world)
            l_r, p_r, G_r, _ = on.UNetOracle(Wt, nfb, force=decl[r]).loss_and_grads(x[s_r], y[s_r], {k: v[s_r] for k, v in masks.items()})
            gs.append(G_r)
            ls.append(l_r)
            ps.append(p_r)
        G_mean = {k: [np.mean([g[k][j] for g in gs], 0) for j in range(len(gs[0][k]))] for k in G_ref}
        worst_l, rel_l, _ = grad_report(Gl, G_mean, "'local' BatchNorm vs independent shards, forced gates: ")
        out = dict(world=world,
                   local_vs_shards_p_err=float(np.abs(pl - ps[0]).max()),
                   local_vs_shards_loss_err=abs(lossl - float(np.mean(ls))),
                   local_vs_shards_grad_rel=rel_l, local_vs_shards_grad_worst=worst_l,
                   p_err=float(np.abs(p - p_ref[sl]).max()), loss_err=abs(loss - loss_ref),
                   grad_cos=cos, grad_rel=rel, grad_worst=worst,
                   # 'local' mode is a different function of the batch: it must NOT match the batch-4 oracle
                   local_p_err=float(np.abs(pl - p_ref[sl]).max()),
                   local_grad_rel=float(np.linalg.norm(fl - fr) / np.linalg.norm(fr)))
        json.dump(out, open(out_path, 'w'))
    parallel.barrier()


if __name__ == '__main__':
    main()
