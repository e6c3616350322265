"""Batch-sharded data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-device code; the path shards naturally over the batch:
every rank draws the SAME global batch from the reference's single RNG stream and trains on its contiguous
slice; the only exchange is ONE all-reduce of the flat 31 MB gradient buffer (+ one 64-byte all-reduce of the
loss/metric sums) per step.  BatchNorm statistics stay local to a rank ('local' mode, standard DP semantics);
moving statistics are averaged across ranks by `sync_moving_stats` before validation / checkpoints.
On CPU (tests) the same functions run over the gloo backend.
"""
import os

import torch
import torch.distributed as dist


def ctypes_void(h):
    import ctypes
    return ctypes.c_void_p(h)


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def forced():
    """DC_DIST_FORCE=1: run the data-parallel code path -- process group, bucketed asynchronous gradient all-reduces issued
    from the weight-gradient stream, metric-sum all-reduce -- even with ONE rank.  On the 1-GPU test box this is the only
    way to put RCCL itself (backend 'nccl', a single-rank communicator) under the step's stream / event choreography."""
    return os.environ.get('DC_DIST_FORCE', '0') == '1'


def exchange_active():
    """True when collectives are to be issued: a process group exists and (more than one rank, or DC_DIST_FORCE=1)."""
    return is_dist() and (world_size() > 1 or forced())


def init_from_env(backend=None):
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); no-op for a single process
    (unless DC_DIST_FORCE=1: a one-rank group, see forced())."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not forced()) or is_dist():
        return rank(), world_size()
    if ws <= 1:
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('RANK', '0')
        if 'MASTER_PORT' not in os.environ:       # a one-rank group needs no agreed port: take a free one
            import socket
            sk = socket.socket()
            sk.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
            sk.close()
    local = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    # RCCL's intra-node transport needs dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0).  The variable is read when the HSA
    # runtime initialises, i.e. at the first GPU call of the process: launchers (bench.py, torchrun wrappers) must export
    # it; setting it here only helps when nothing has touched the GPU yet.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local % torch.cuda.device_count())
    # DC_DIST_BACKEND=gloo lets several ranks share ONE GPU (functional testing of the N>1 path on a 1-GPU box)
    backend = backend or os.environ.get('DC_DIST_BACKEND') or ('nccl' if use_cuda else 'gloo')
    dist.init_process_group(backend)
    import atexit
    atexit.register(shutdown)          # a process group alive at interpreter exit can abort in its threads' destructors
    return rank(), world_size()


def shutdown():
    """Tear the process group down (idempotent; registered at exit by init_from_env): without it the backend's worker
    threads are destroyed while still joinable when the interpreter exits -- 'terminate called without an active exception',
    SIGABRT after a run that had finished fine."""
    if _NATIVE:
        try:
            from ._lib import lib
            for h in _NATIVE.values():
                if h:
                    lib().cdll.dc_comm_destroy(ctypes_void(h))
        except Exception:
            pass
        _NATIVE.clear()
    if is_dist():
        try:
            dist.destroy_process_group()
        except Exception:
            pass


_NATIVE = {}          # device index -> dc_comm_* handle (RCCL through the C ABI) of this process


def native_comm(device):
    """The gradient exchange's RCCL communicator behind the C ABI (include/dcunet.h dc_comm_*, csrc/comm.cpp), made on first use:
    rank 0 draws the unique id, the others receive it through the process group that rendezvoused the ranks.  None -- the
    exchange then goes through torch.distributed -- when no exchange is active, when the group's backend is not RCCL (gloo:
    the CPU tests and the ranks-share-one-GPU functional mode, where RCCL refuses two ranks on one device) or under DC_COMM=torch."""
    if not exchange_active() or dist.get_backend() != 'nccl' or os.environ.get('DC_COMM', 'rccl') == 'torch':
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    h = _NATIVE.get(idx, 0)
    if h == 0:
        import ctypes
        from ._lib import lib
        L = lib()
        nbytes = 128                      # DC_COMM_ID_BYTES
        buf = ctypes.create_string_buffer(nbytes)
        ok, why = 1, ''
        # can THIS rank load RCCL at all?  (an empty group start / end resolves the library; agreed on by all ranks BEFORE any of
        # them enters the collective communicator set-up, which would otherwise wait forever for the one that cannot)
        if L.cdll.dc_comm_group_start() != 0 or L.cdll.dc_comm_group_end() != 0:
            ok, why = 0, L.cdll.dc_last_error().decode()
        pre = torch.tensor([ok], dtype=torch.int32, device=torch.device('cuda', idx))
        if world_size() > 1:
            dist.all_reduce(pre, op=dist.ReduceOp.MIN)
        can = int(pre.item()) == 1
        if can and rank() == 0 and L.cdll.dc_comm_unique_id(buf) != 0:
            ok, why = 0, L.cdll.dc_last_error().decode()
        box = [buf.raw, ok if can else 0]
        if world_size() > 1:
            dist.broadcast_object_list(box, 0)
        comm = ctypes.c_void_p()
        if box[1]:
            ident = ctypes.create_string_buffer(box[0], nbytes)
            with torch.cuda.device(idx):
                if L.cdll.dc_comm_init_rank(ctypes.byref(comm), ident, world_size(), rank()) != 0:
                    ok, why = 0, L.cdll.dc_last_error().decode()
        else:
            ok = 0
        # every rank takes the SAME path: one that cannot bring its communicator up sends all of them to torch.distributed
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device('cuda', idx))
        if world_size() > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm.value:
                L.cdll.dc_comm_destroy(comm)
            import warnings
            warnings.warn('dc_comm_* (RCCL through the C ABI) is not available on every rank%s: the gradient exchange goes through '
                          'torch.distributed' % ((': ' + why) if why else ''))
            _NATIVE[idx] = h = None
        else:
            h = _NATIVE[idx] = comm.value
    return h


def native_comm_active(device):
    return native_comm(device) is not None


def shard_slice(global_batch, r=None, ws=None):
    """Contiguous slice [r*B/G, (r+1)*B/G) of the global batch owned by rank r."""
    r = rank() if r is None else r
    ws = world_size() if ws is None else ws
    if ws == 1:
        return slice(0, global_batch)
    if global_batch % ws:
        raise ValueError('global batch %d is not divisible by world size %d' % (global_batch, ws))
    per = global_batch // ws
    return slice(r * per, (r + 1) * per)


def all_reduce_sum(t):
    """In-place sum over the ranks, ordered on the CURRENT stream of t's device: through dc_comm_all_reduce_sum[_f64] (RCCL behind
    the C ABI) for contiguous fp32 / fp64 device tensors when the native communicator is up, else torch.distributed."""
    if not exchange_active():
        return t
    if t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64) and t.numel() > 0:
        comm = native_comm(t.device)
        if comm is not None:
            from ._lib import lib
            L = lib()
            with torch.cuda.device(t.device):
                st = torch.cuda.current_stream(t.device).cuda_stream
                (L.dc_comm_all_reduce_sum if t.dtype == torch.float32 else L.dc_comm_all_reduce_sum_f64)(comm, t.data_ptr(), t.numel(), st)
            return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max(t):
    if exchange_active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_reduce_sum_host(a):
    """Sum a host (numpy float64) array over the ranks and return it: through a device tensor under RCCL, a CPU tensor under
    gloo.  Used for small result tables (per-item validation scores: each item is owned by one rank, the others add 0.0)."""
    import numpy as np
    if not exchange_active():
        return a
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().reshape(np.shape(a))


def all_gather_floats(value, device=None):
    """One float per rank -> list over ranks (bench.py: per-rank timings)."""
    if not exchange_active():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if exchange_active():
        dist.barrier()


def sync_moving_stats(sflat):
    """Average the BatchNorm moving statistics over ranks (they drift apart under local-BN data parallelism): the sum is
    the collective, the 1/G a libdcunet launch (no torch arithmetic on device tensors in the product path; a CPU tensor --
    the gloo plumbing test -- is scaled in place)."""
    if exchange_active():
        all_reduce_sum(sflat)
        if sflat.is_cuda:
            from ._lib import lib
            with torch.cuda.device(sflat.device):
                lib().dc_scale_flat(sflat.data_ptr(), sflat.numel(), 1.0 / world_size(),
                                    torch.cuda.current_stream(sflat.device).cuda_stream)
        else:
            sflat.div_(world_size())
    return sflat


def broadcast_params(*tensors, src=0):
    if exchange_active():
        for t in tensors:
            dist.broadcast(t, src)


def broadcast_numpy_rng(src=0):
    """Make numpy's GLOBAL RNG (the stream the reference's _batch_gen draws from, unet_2d_summary.py:434-530) identical
    on every rank: rank `src`'s state is broadcast, so a seeded multi-GPU run samples exactly the batches the same
    seeded single-GPU run would, whatever the other ranks did with their RNG before."""
    if not (is_dist() and world_size() > 1):
        return
    import numpy as np
    box = [np.random.get_state() if rank() == src else None]
    dist.broadcast_object_list(box, src)
    np.random.set_state(box[0])


GOLDEN64 = 0x9E3779B97F4A7C15        # the multiplier of dc_hash32 (csrc/common.h)


def shard_drop_seed(seed, local_elems, r=None):
    """Dropout seed of rank r's shard such that its element e draws the bits a single device would have drawn for
    element r*local_elems + e of the global batch: dc_hash32 mixes idx*GOLDEN64 + seed, so an index offset is a seed
    offset (mod 2^64).  Without it every rank would apply the same masks to its shard."""
    r = rank() if r is None else r
    if seed == 0 or r == 0:
        return seed
    return (seed + r * int(local_elems) * GOLDEN64) & 0xFFFFFFFFFFFFFFFF
