"""Every surviving A/B switch of the product (README.md "Switches") runs the whole-network train-step parity test under its
alternate setting, in a fresh process (the switches are read once at import / engine construction):
tests/test_engine_gpu.py::test_train_forward_backward_matches_oracle at [2-32-32-32] and [2-64-64-32] (the sizes at
which the role-split kernel, BN-on-load and dz-on-load all engage) -- same bounds as under the defaults."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

HERE = os.path.dirname(os.path.abspath(__file__))
SWITCHES = [('DC_MFMA', 'f32'), ('DC_STREAMS', '1'), ('DC_IGEMM_PP', '0'), ('DC_IGEMM_PP', '2'), ('DC_BNIN', '0'), ('DC_DZIN', '0'),
            ('DC_DZIN', 'all'), ('DC_BN_MODE', 'sync'), ('DC_INFER_GUARD', '1'), ('DC_AR_BUCKETS', '1'), ('DC_COMM', 'torch'), ('DC_TAPES', '0'), ('DC_EVENT_SYSTEM_FENCE', '1'),
            ('DC_TAIL_MAIN', '0'), ('DC_STATS_FOLD_MIN', '0'), ('DC_STATS_FOLD_MIN', '4')]


@pytest.mark.parametrize('name,value', SWITCHES, ids=['%s=%s' % s for s in SWITCHES])
def test_train_step_parity_under_switch(name, value):
    env = dict(os.environ)
    env[name] = value
    if name in ('DC_AR_BUCKETS', 'DC_COMM'):          # the exchange path: a one-rank RCCL group (one blocking all-reduce / torch.distributed's)
        env.update(DC_DIST_FORCE='1', DC_DIST_BACKEND='nccl', MASTER_ADDR='127.0.0.1', MASTER_PORT='29571', HSA_ENABLE_IPC_MODE_LEGACY='0')
    sel = 'test_train_forward_backward_matches_oracle and f16x3 and (2-32-32-32 or 2-64-64-32)'
    files = [os.path.join(HERE, 'test_engine_gpu.py')]
    if name == 'DC_INFER_GUARD':         # an inference switch: the forward-parity test is the one that exercises it
        sel = 'test_forward_inference_matches_oracle and f16x3'
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-k', sel] + files, env=env, capture_output=True,
                       text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-500:]
