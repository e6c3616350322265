"""Batch-sharded data parallelism on the GPU.  The test box has ONE GPU: ranks share it over gloo
(DC_DIST_BACKEND=gloo; the collective calls are the ones RCCL serves on a multi-GPU node)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run_ranks(script, out, port, backend='gloo'):
    env = dict(os.environ, DC_DIST_BACKEND=backend, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(HERE, script), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.load(open(out))


def test_local_mode_step_equals_independent_shards(tmp_path):
    res = _run_ranks('_dp_step_worker.py', str(tmp_path / 'res.json'), 29541)
    print(res)
    assert res['world'] == 2
    assert res['loss_err'] < 1e-4 and res['grad_rel'] < 1e-4 and res['grad_worst'] < 1e-4, res
    # three overlapped all-reduces of contiguous ranges == one all-reduce of the flat gradient, bit for bit
    assert res['buckets_bitwise_grad'] and res['buckets_bitwise_params'], res
    assert res['seeds_distinct'] and res['rng_same'], res


def test_validation_logs_identical_on_every_rank(tmp_path):
    """_ValidationMetricsCB under data parallelism (round 5): the 6 n validation items are dealt round-robin to the ranks (inference
    is 'replicas only'), every rank forwards and scores its share, the per-item scores are summed over the ranks (each item is owned
    by exactly one rank) -- so both ranks write the SAME val_nf_* into their logs, equal to what one rank computes alone.  2 ranks on
    one GPU over gloo; reference: /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:62-120."""
    res = _run_ranks('_val_logs_worker.py', str(tmp_path / 'val.json'), 29571)
    assert res['world'] == 2 and res['nonzero']
    a, b = res['logs']
    assert a == b and set(a) == {'val_nf_f1_mean', 'val_nf_f1_median', 'val_nf_f1_min', 'val_nf_f1_adj', 'val_nf_prec', 'val_nf_reca'}
    assert a['val_nf_f1_mean'] == res['alone']['f1_mean'] and a['val_nf_prec'] == res['alone']['prec'] and a['val_nf_reca'] == res['alone']['reca']


two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL between two devices over xGMI')


@two_gpus
def test_local_mode_step_equals_independent_shards_over_rccl(tmp_path):
    """The same worker, one rank per DEVICE over nccl (= RCCL): the first multi-GPU box that runs the suite validates the
    exchange itself (the 1-GPU box skips it)."""
    res = _run_ranks('_dp_step_worker.py', str(tmp_path / 'res.json'), 29561, backend='nccl')
    assert res['world'] == 2
    assert res['loss_err'] < 1e-4 and res['grad_rel'] < 1e-4 and res['grad_worst'] < 1e-4, res
    assert res['buckets_bitwise_grad'] and res['buckets_bitwise_params'], res
    assert res['seeds_distinct'] and res['rng_same'], res


@two_gpus
def test_bench_two_gpus_over_rccl():
    """`python bench.py --gpus 2` on two devices: the line must say so (rccl_ranks == 2, shared_gpus false)."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DC_DIST_BACKEND'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['config']['shared_gpus'] is False
    assert out['config']['dist_backend'] == 'nccl' and out['allreduce_exposed_ms'] < out['ms_per_step']


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs) starts its own rank processes and
    prints ONE JSON line for the 2-rank job."""
    env = dict(os.environ, DC_DIST_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['scaling'] == 'weak'
    assert out['value'] > 0 and out['allreduce_ms'] > 0 and out['allreduce_bytes'] == 4 * 7766402
    assert out['config']['shared_gpus'] is True and np.isfinite(out['config']['loss'])


def test_bench_one_rank_over_rccl_bucketed_async_path():
    """RCCL itself under the step's stream / event choreography, every round, on the 1-GPU box: `bench.py --gpus 1` with
    DC_DIST_FORCE=1 builds a ONE-rank 'nccl' process group and runs the data-parallel step -- three bucketed asynchronous
    all-reduces issued from the weight-gradient stream while the encoder's backward runs, metric-sum all-reduce, 1/G in Adam."""
    env = dict(os.environ, DC_DIST_FORCE='1', DC_DIST_BACKEND='nccl', MASTER_ADDR='127.0.0.1', MASTER_PORT='29553',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['rccl_ranks'] == 1 and out['config']['dist_backend'] == 'nccl'
    assert out['allreduce_buckets'] == 3 and out['allreduce_bytes'] == 4 * 7766402
    assert out['allreduce_ms'] >= 0 and out['allreduce_exposed_ms'] >= 0 and len(out['per_rank_images_per_s']) == 1
    assert out['value'] > 100 and np.isfinite(out['config']['loss'])
    print(lines[0])


def test_bench_watchdog_names_the_hung_phase():
    """A rank that stops making progress exits 17 and says where (DC_BENCH_TEST_HANG: the rank goes to sleep at timed step 2;
    per-step limit 3 s instead of 30)."""
    env = dict(os.environ, DC_BENCH_WATCHDOG_S='3', DC_BENCH_TEST_HANG='2')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '1', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert "watchdog: rank 0 made no progress in phase 'timed step 2'" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_example_train_under_torch_distributed_run(tmp_path):
    """The documented multi-GPU command: `python -m torch.distributed.run --nproc-per-node 2 examples/neurons/unet2ds_nf.py
    train <name>`.  Both ranks replay the same sampling stream on their half of every batch of 20, validation runs on
    each, rank 0 alone writes the metrics CSV and the checkpoints, and the loss goes down (2 epochs of 5 steps here)."""
    sys.path.insert(0, HERE)
    from _nf_dirs import make_neurofinder_dir
    home = str(tmp_path / 'home')
    name = 'neurofinder.01.00'
    root = make_neurofinder_dir('%s/.deep-calcium/datasets/neurons_nf' % home, name, seed=5)
    cp = str(tmp_path / 'cp')
    env = dict(os.environ, DC_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', HOME=home)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29547', os.path.join(ROOT, 'examples', 'neurons', 'unet2ds_nf.py'), 'train', name, '-c', cp,
           '--nb_steps', '5', '--nb_epochs', '2']        # (over gloo on ONE shared GPU every 31 MB gradient exchange costs ~0.5 s)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert os.path.exists(root + '/dataset.hdf5')
    files = os.listdir(cp)
    csvs = [f for f in files if f.endswith('.csv')]
    assert len(csvs) == 1, files                                    # rank 0 only
    assert len([f for f in files if '_model_' in f and f.endswith('.hdf5')]) == 2, files
    rows = open(os.path.join(cp, csvs[0])).read().strip().splitlines()
    cols = rows[0].split(',')
    first, last = [dict(zip(cols, r_.split(','))) for r_ in (rows[1], rows[-1])]
    assert len(rows) == 3 and float(last['loss']) < float(first['loss'])


def test_shard_dropout_seed_reproduces_single_device_masks():
    """RNG dropout under data parallelism: rank r's shard with the offset seed gets the keep-bits ONE device draws for the
    same elements of the global batch (common.h dc_hash32: an element-index offset is a seed offset)."""
    from deep_calcium_amd import parallel
    from deep_calcium_amd._lib import lib
    L = lib()
    N, h, w, C = 4, 8, 8, 16
    pix = N * h * w
    dev = 'cuda'
    z = torch.ones(N, h, w, C, device=dev)
    mean, beta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    invstd, gamma = torch.ones(C, device=dev), torch.ones(C, device=dev)
    seed, keep = 0x5eed1234abcd, 0.5
    st = torch.cuda.current_stream().cuda_stream

    def run(zz, s):
        out = torch.empty_like(zz)
        L.dc_bn_relu_drop_fwd(zz.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, keep, s,
                              out.data_ptr(), C, zz.numel() // C, C, 0.0, None, st)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    full = run(z, seed)
    assert 0.3 < (full > 0).mean() < 0.7
    world = 2
    per = N // world
    for r in range(world):
        part = run(z[r * per:(r + 1) * per].contiguous(), parallel.shard_drop_seed(seed, per * h * w * C, r))
        assert np.array_equal(part, full[r * per:(r + 1) * per])
    assert not np.array_equal(run(z[per:].contiguous(), seed), full[per:])      # without the offset the masks repeat


def test_bench_default_line_schema():
    """`python bench.py` (N = 1): ONE JSON line on stdout with every field of the driver's contract, the roofline object of
    the dominant kernel and the CPU baseline."""
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['n_gpus'] == 1 and out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['unit'] == 'images/s' and out['higher_is_better'] is True and out['vs_baseline'] is None
    assert out['dtype'] == 'f32 (fp16x3 split, fp32 accumulate)' and out['data'] == 'synthetic' and 'workload' in out['config'] and 'model' not in out['config']
    assert abs(out['value'] - 16 * 1000.0 / out['ms_per_step']) < 0.01 * out['value']
    rf = out['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and 0.05 < rf['frac'] < 0.25
    assert rf['traffic'] is None or rf['traffic'] > 1e8
    # 2 instrumented steps x the 12 launches per step that dispatch igemm_pp_kernel<2,2,0,false>: the 7 training-forward convolutions
    # of levels 3-4 with > 32 output columns and >= 64 input channels + the 5 plain data gradients (round 5: the 7 forward
    # convolutions of levels 1-2 merge their BatchNorm partials per workgroup: instantiation <2,2,3,false>, tallied separately)
    # round 6: the weight gradients are timed too -- 13 launches per step of wgrad_f16x3_kernel<3,3,1,1,16,4,2,2,1,false,false> (every
    # conv3x3 layer with Cin, Cout > 32), the trace's #1 symbol; the line names whichever symbol took more time on this box
    assert (rf['kernel'], rf['launches']) in (('igemm_pp_kernel<2,2,0,false>', 2 * 12),
                                              ('wgrad_f16x3_kernel<3,3,1,1,16,4,2,2,1,false,false>', 2 * 13)), rf
    assert 1.0e8 < rf['algorithmic_bytes_per_launch'] < 3.3e8
    cb = out['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['unit'] == 'images/s' and cb['value'] > 0 and cb['cores'] >= 1
