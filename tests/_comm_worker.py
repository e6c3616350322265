"""Worker of tests/test_comm_gpu.py (one process per rank, backend nccl): seeded data-parallel training through
Model.train_on_device_batch with the gradient exchange (a) through the C ABI's dc_comm_* on a launch tape, (b) the same launch by
launch, (c) through torch.distributed (DC_COMM=torch: the tape is cut at every bucket hand-over); writes the digests of the
parameters / Adam state after the last step and what the backward tape holds."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import parallel                      # noqa: E402
from deep_calcium_amd.model import Model, Adam             # noqa: E402
from deep_calcium_amd._lib import MARK                     # noqa: E402


def digest(eng):
    h = hashlib.sha256()
    for t in (eng.pflat, eng.mflat, eng.vflat, eng.sflat):
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()


def main():
    out_path, steps = sys.argv[1], int(sys.argv[2])
    modes = sys.argv[3].split(',')
    rank, world = parallel.init_from_env()
    H, nfb, B = 64, 16, 4
    rs = np.random.RandomState(31 + rank)
    data = [(torch.from_numpy(rs.standard_normal((B, H, H)).astype(np.float32)).cuda(),
             torch.from_numpy((rs.random_sample((B, H, H)) < 0.15).astype(np.uint8)).cuda()) for _ in range(2)]
    res = dict(world=world, rank=rank, backend=torch.distributed.get_backend())
    for mode in modes:
        os.environ['DC_COMM'] = 'torch' if mode == 'torch' else 'rccl'
        m = Model((H, H), nfb)
        m.compile(Adam(0.002), 'binary_crossentropy')
        m.engine.use_tapes = mode != 'native_untaped'
        hist = [m.train_on_device_batch(*data[s % 2]) for s in range(steps)]
        torch.cuda.synchronize()
        eng = m.engine
        info = dict(digest=digest(eng), loss=[float(h[0]) for h in hist], replays=eng.tape_replays,
                    native=parallel.native_comm(eng.device) is not None)
        for key, ent in eng._tapes.items():
            if key[0] == 'bwd':
                names = [n for n, _ in ent['ops']]
                info.update(bwd_marks=names.count(MARK), bwd_allreduces=names.count('dc_comm_all_reduce_sum'),
                            bwd_taped=ent['tape'] is not None, bwd_segments=len(ent['tape'].segments) if ent['tape'] is not None else 0)
        res[mode] = info
    allres = [None] * world
    torch.distributed.all_gather_object(allres, res)
    if rank == 0:
        json.dump(allres, open(out_path, 'w'))
    parallel.barrier()


if __name__ == '__main__':
    main()
