"""The N>1 path on CPU: world_size-2 gloo processes exercise the data-parallel plumbing of
deep_calcium_amd.parallel (batch sharding, flat-gradient all-reduce + 1/G scaling, metric-sum all-reduce,
moving-statistics averaging).  The per-rank gradients come from the oracle (the HIP kernels need a GPU); what is
checked is that shard -> all-reduce -> scale reproduces the gradient of the GLOBAL-batch mean loss with local BN
statistics, i.e. the semantics of SURVEY 8(e) 'local' mode."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_numpy as on

NFB, N, H, W = 4, 4, 16, 16


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(G):
    return np.concatenate([g.ravel() for name in G for g in G[name]])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from deep_calcium_amd import parallel
    from deep_calcium_amd.model import metrics_from_sums
    r, w = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and parallel.world_size() == world
    # over gloo the exchange stays with torch.distributed: the RCCL communicator behind the C ABI (dc_comm_*) is for backend nccl only
    assert parallel.native_comm('cuda:0') is None and not parallel.native_comm_active('cuda:0')
    x, y = on.synthetic_batch(N, H, W)                      # every rank draws the SAME global batch
    masks = on.make_drop_masks(NFB, N, H, W)
    sl = parallel.shard_slice(N)
    assert (sl.start, sl.stop) == (rank * N // world, (rank + 1) * N // world)
    Wt = on.init_weights(NFB, randomize_bn=True, dtype=np.float64)
    loss, p, G, _ = on.UNetOracle(Wt, NFB).loss_and_grads(x[sl], y[sl], {k: v[sl] for k, v in masks.items()})
    g = torch.from_numpy(_flat(G))
    parallel.all_reduce_sum(g)
    g /= world                                              # what dc_adam_step_flat's gscale = 1/G does
    yt = y[sl].astype(np.float64)
    pr = np.round(p)
    sums = torch.tensor([loss * p.size, (pr * yt).sum(), pr.sum(), np.clip(yt - pr, 0, 1).sum(), yt.sum(),
                         (yt * p).sum(), (p ** 2).sum(), (yt ** 2).sum()], dtype=torch.float64)
    parallel.all_reduce_sum(sums)
    stats = torch.full((6,), float(rank))
    parallel.sync_moving_stats(stats)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    parallel.all_reduce_max(t)
    parallel.barrier()
    if rank == 0:
        m = metrics_from_sums(sums.numpy(), float(N * H * W))
        np.savez(os.path.join(out_dir, 'r0.npz'), g=g.numpy(), loss=m['loss'], f1=m['F1'], stats=stats.numpy(), tmax=t.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(str(tmp_path / 'r0.npz'))
    # reference: the two shards processed independently (local BN), gradients averaged
    x, y = on.synthetic_batch(N, H, W)
    masks = on.make_drop_masks(NFB, N, H, W)
    Wt = on.init_weights(NFB, randomize_bn=True, dtype=np.float64)
    gs, losses, ps = [], [], []
    for r in range(world):
        sl = slice(r * N // world, (r + 1) * N // world)
        loss, p, G, _ = on.UNetOracle(Wt, NFB).loss_and_grads(x[sl], y[sl], {k: v[sl] for k, v in masks.items()})
        gs.append(_flat(G))
        losses.append(loss)
        ps.append(p)
    assert np.abs(z['g'] - np.mean(gs, 0)).max() < 1e-14
    assert abs(float(z['loss']) - np.mean(losses)) < 1e-12
    assert abs(float(z['f1']) - on.keras_metrics(y, np.concatenate(ps))['F1']) < 1e-12
    assert np.allclose(z['stats'], 0.5) and float(z['tmax'][0]) == 2.0


def test_shard_slice_contract():
    from deep_calcium_amd import parallel
    assert parallel.shard_slice(128, 3, 8) == slice(48, 64)
    assert parallel.shard_slice(16, 0, 1) == slice(0, 16)
    with pytest.raises(ValueError):
        parallel.shard_slice(10, 0, 4)


def _dataset_worker(rank, world, port, ds_dir, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from deep_calcium_amd import parallel
    from deep_calcium_amd.nf_datasets import nf_load_hdf5
    parallel.init_from_env(backend='gloo')
    paths = nf_load_hdf5('neurofinder.01.00,neurofinder.02.00', datasets_dir=ds_dir)      # EVERY rank calls it
    leftovers = [f for n in ('neurofinder.01.00', 'neurofinder.02.00') for f in os.listdir('%s/%s' % (ds_dir, n)) if 'partial' in f]
    with open(os.path.join(out_dir, 'r%d.txt' % rank), 'w') as fp:
        fp.write('\n'.join(paths + ['%d' % os.path.getsize(p) for p in paths] + ['leftovers=%d' % len(leftovers)]))
    parallel.barrier()
    dist.destroy_process_group()


def test_dataset_build_under_two_ranks(tmp_path):
    """fit() under data parallelism: every rank calls nf_load_hdf5; rank 0 alone builds dataset.hdf5 from the unpacked
    directory, the others wait at the barrier inside and then see the finished file (never a half-extracted directory or
    a truncated file), and the call returns on all ranks (no rank-0-only call sites with a barrier inside)."""
    from _nf_dirs import make_neurofinder_dir
    from deep_calcium_amd import hdf5_min
    ds_dir = str(tmp_path / 'neurons_nf')
    for k, name in enumerate(('neurofinder.01.00', 'neurofinder.02.00')):
        make_neurofinder_dir(ds_dir, name, hw=(64, 64), neurons=6, frames=3, seed=3 + k)
    world = 2
    mp.spawn(_dataset_worker, args=(world, _free_port(), ds_dir, str(tmp_path)), nprocs=world, join=True)
    a, b = [open(str(tmp_path / ('r%d.txt' % r))).read() for r in range(world)]
    assert a == b and a.endswith('leftovers=0')
    for line in a.splitlines()[:2]:
        f = hdf5_min.File(line)
        assert f['series/raw'].shape == (3, 64, 64) and f['masks/raw'].shape == (6, 64, 64)


def test_bench_watchdog_exits_17_and_names_the_phase():
    """bench.py's per-rank hang detector on its own (no GPU): a phase that outlives its deadline ends the process with
    exit code 17 and one line naming rank and phase; a phase that is ticked in time does not."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; wd = bench.Watchdog(3)\n"
            "for i in range(4):\n"
            "    wd.tick('step %%d' %% i); time.sleep(0.2)\n"
            "wd.phase('stuck collective'); time.sleep(30)\n" % root)
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DC_BENCH_WATCHDOG_S='1.0'), capture_output=True, text=True, timeout=60)
    assert r.returncode == 17, (r.returncode, r.stderr)
    assert "rank 3 made no progress in phase 'stuck collective'" in r.stderr
    ok = subprocess.run([sys.executable, '-c', code.replace("time.sleep(30)", "wd.done(); time.sleep(2.5)")],
                        env=dict(os.environ, DC_BENCH_WATCHDOG_S='1.0'), capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0, ok.stderr
