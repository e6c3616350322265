"""Data-parallel BatchNorm modes on the GPU: two ranks share the one GPU of the test box over gloo
(DC_DIST_BACKEND), each trains on half of a batch of 4; 'sync' mode must equal the oracle's single-device batch-4 step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('backend', ['gloo', 'gloo-dzin', pytest.param('nccl', marks=pytest.mark.skipif(
    torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL between two devices'))])
def test_sync_bn_two_ranks_equal_one_device_batch(tmp_path, backend):
    out = str(tmp_path / 'res.json')
    extra = {}
    if backend == 'gloo-dzin':      # nfb 32 + DC_DZIN=all: dz on load / the joint kernel with the ALL-REDUCED (dgamma, dbeta) and the
        backend, extra = 'gloo', dict(DC_DZIN='all', DC_TEST_NFB='32')      # global count in dc_bn_bwd_finalize_dzin's table
    env = dict(os.environ, DC_DIST_BACKEND=backend, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', **extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(29533 + len(extra) + (backend == 'nccl')), os.path.join(HERE, '_sync_bn_worker.py'), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res['world'] == 2
    assert res['p_err'] < 1e-4 and res['loss_err'] < 1e-4, res
    # gradients with the oracle routed through the ranks' own ReLU gates / pool indices: rounding only (tests/_forced.py)
    assert res['grad_rel'] < 1e-4 and res['grad_worst'] < 1e-4, res
    assert res['local_p_err'] > 1e-3 and res['local_grad_rel'] > 0.05, res      # the two modes really differ
    # 'local' mode == the oracle run as independent shards with averaged gradients
    assert res['local_vs_shards_p_err'] < 1e-4 and res['local_vs_shards_loss_err'] < 1e-4, res
    assert res['local_vs_shards_grad_rel'] < 1e-4 and res['local_vs_shards_grad_worst'] < 1e-4, res
    print(res)
