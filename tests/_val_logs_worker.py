"""Worker of tests/test_dp_gpu.py::test_validation_logs_identical_on_every_rank (one process per rank): the validation callback
with its items dealt over the ranks and the per-item scores summed back must write the SAME logs on every rank -- and the logs a
single rank computes alone (same weights: seed-initialised, a few identical training steps)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import parallel, unet_hip, Adam                     # noqa: E402
from deep_calcium_amd.unet2ds import _ValidationMetricsCB                 # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = parallel.init_from_env()
    rs = np.random.RandomState(3)
    S, M = [], []
    for k in range(3):
        hw = (72, 80 + 8 * k)
        s = (rs.random_sample(hw) * 0.3).astype(np.float32)
        m = np.zeros(hw)
        for _ in range(12):
            cy, cx = rs.randint(6, hw[0] - 6), rs.randint(6, hw[1] - 6)
            m[cy - 3:cy + 4, cx - 3:cx + 4] = 1
            s[cy - 3:cy + 4, cx - 3:cx + 4] += 2.0
        S.append((s - s.mean()) / s.std())
        M.append(m)
    yc = [(s.shape[0] - int(s.shape[0] * 0.25), s.shape[0]) for s in S]
    model = unet_hip((32, 32), nb_filters_base=8)
    model.compile(Adam(0.002))
    for it in range(10):                      # every rank trains on the SAME global batches: identical weights, averaged stats
        k = it % 3
        y0, x0 = 5 * it % 40, 7 * it % 48
        xb = np.stack([S[k][y0:y0 + 32, x0:x0 + 32]] * 4).astype(np.float32)
        yb = np.stack([M[k][y0:y0 + 32, x0:x0 + 32]] * 4).astype(np.uint8)
        model.train_on_batch(xb, yb)
    model_val = unet_hip((112, 112), nb_filters_base=8)
    cb = _ValidationMetricsCB(model_val, S, M, ['a', 'b', 'c'], yc)
    cb.set_model(model)
    logs = {}
    cb.on_epoch_end(2, logs)
    # what ONE rank computes alone on the same weights (the reference's sequence: predict one by one, round, score)
    scores = cb._score_through_predict(len(cb.S_summ))
    alone = dict(f1_mean=float(np.mean(scores[:, 2]) + 2e-4), prec=float(np.mean(scores[:, 0])), reca=float(np.mean(scores[:, 1])))
    box = [None] * world
    torch.distributed.all_gather_object(box, dict((k, float(v)) for k, v in logs.items()))
    if rank == 0:
        json.dump(dict(world=world, logs=box, alone=alone, nonzero=bool(scores[:, 2].max() > 0)), open(out_path, 'w'))
    parallel.barrier()


if __name__ == '__main__':
    main()
