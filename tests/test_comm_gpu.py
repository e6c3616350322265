"""The gradient exchange through the C ABI (include/dcunet.h dc_comm_*, csrc/comm.cpp: RCCL resolved with dlopen).
On the 1-GPU box RCCL runs with ONE rank (DC_DIST_FORCE=1: the all-reduce is the identity, everything around it -- communicator
set-up through the process group's store, the collective stream, the fenced hand-over events, the tape -- is the real thing); the
two-rank twins run whenever two devices are visible.  The reference has no multi-device code; the step being exchanged is
model.fit_generator -> train_on_batch (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:429-430)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
HERE = os.path.dirname(os.path.abspath(__file__))
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL between two devices over xGMI')


def _run(out, nproc, port, steps, modes, extra_env=None):
    env = dict(os.environ, DC_DIST_BACKEND='nccl', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env or {})
    if nproc == 1:
        env.update(DC_DIST_FORCE='1', MASTER_PORT=str(port))
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(HERE, '_comm_worker.py'), out, str(steps), modes]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(HERE, '_comm_worker.py'), out, str(steps), modes]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


def _check(allres):
    for res in allres:
        a, b, c = res['native'], res['native_untaped'], res['torch']
        assert res['backend'] == 'nccl' and a['native'] and b['native'] and not c['native']
        # the backward of a data-parallel step is ONE tape segment holding the three all-reduces; the torch.distributed path is
        # cut at the two bucket hand-overs (three segments, the collectives issued from Python in between)
        assert a['bwd_taped'] and a['bwd_marks'] == 0 and a['bwd_allreduces'] == 3 and a['bwd_segments'] == 1, a
        assert c['bwd_taped'] and c['bwd_marks'] == 2 and c['bwd_segments'] == 3 and c['bwd_allreduces'] == 0, c
        assert a['replays'] > 0 and b['replays'] == 0
        assert a['digest'] == b['digest'] == c['digest'], (a['loss'], b['loss'], c['loss'])     # parameters, Adam state, moving statistics: bit for bit
        assert a['loss'] == b['loss'] == c['loss']
    assert len(set(r['native']['digest'] for r in allres)) == 1             # every rank holds the same model


def test_one_rank_rccl_step_through_the_abi_taped_end_to_end(tmp_path):
    allres = _run(str(tmp_path / 'c.json'), 1, 29581, 7, 'native,native_untaped,torch')
    assert allres[0]['world'] == 1
    _check(allres)


@two_gpus
def test_two_rank_rccl_step_through_the_abi(tmp_path):
    allres = _run(str(tmp_path / 'c.json'), 2, 29583, 7, 'native,native_untaped,torch')
    assert allres[0]['world'] == 2
    _check(allres)


@two_gpus
def test_gradient_handover_to_rccl_default_events_against_fully_fenced_run(tmp_path):
    """Round 5 dropped the system-scope fence from the stream-ordering events (hipEventDisableSystemFence); the events in front of
    and behind a collective keep it (dc_event_create_fenced: a peer GPU reads gflat over xGMI).  With EVERY event fenced
    (DC_EVENT_SYSTEM_FENCE=1) the same seeded 2-rank run must land on the same bits -- a stale gradient element anywhere in 12 steps
    changes the digest."""
    a = _run(str(tmp_path / 'a.json'), 2, 29585, 12, 'native')
    b = _run(str(tmp_path / 'b.json'), 2, 29587, 12, 'native', {'DC_EVENT_SYSTEM_FENCE': '1'})
    assert [r['native']['digest'] for r in a] == [r['native']['digest'] for r in b]
