#!/usr/bin/env python
"""bench.py -- 512x512 summary images/sec for one UNet2DS train step (fwd + BCE + bwd + Adam) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: bench.py starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = BASELINE.json configs[2] per GPU (batch 16 of 512x512, fp32), i.e. configs[3] (global batch 128) at
8 GPUs: weak scaling, one process per GPU, gradients all-reduced over RCCL.  Inputs are synthetic
(x ~ N(0,1), y ~ Bernoulli(0.126)), random-init weights, resident in HBM before the timed region.
Rank 0 prints ONE JSON line: metric/value (whole-job images/s) + `roofline` for the dominant kernel
(measured live with HIP events on the launch stream) + `cpu_baseline` (oracle port timed on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
BATCH_PER_GPU = 16
H = W = 512
NFB = 32


PEAK_FP16_MFMA_TFLOPS = 2500.0    # same guide: dense fp16/bf16 matrix peak (the 2:1-sparse figure is not used)
DTYPE_LABEL = {'f16x3': 'f32 (fp16x3 split, fp32 accumulate)', 'f32': 'f32'}


def synthetic_batch(N, Hh, Ww, seed_x=865, seed_y=866, pos_rate=0.126):
    """SURVEY 8(d) synthetic inputs: x ~ N(0,1) (the real pipeline feeds zero-mean / unit-std summary images,
    /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:239), y ~ Bernoulli(0.126)."""
    x = np.random.RandomState(seed_x).standard_normal((N, Hh, Ww)).astype(np.float32)
    y = (np.random.RandomState(seed_y).random_sample((N, Hh, Ww)) < pos_rate).astype(np.uint8)
    return x, y


class Watchdog(object):
    """Per-rank hang detector: every phase of the run has a deadline; when one passes, the rank says which phase of which
    rank hung and exits non-zero (a hard exit of THIS process -- never a re-exec), so that a launcher (bench.py's own
    or torch.distributed.run) tears the job down instead of sitting in a collective until the driver's limit.
    DC_BENCH_WATCHDOG_S = seconds allowed per train step (default 30; 0 disables)."""

    def __init__(self, rank):
        import threading
        self.rank, self.phase_name, self.deadline = rank, 'start', None
        self.step_s = float(os.environ.get('DC_BENCH_WATCHDOG_S', '30'))
        if self.step_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def phase(self, name, limit_s=None):
        self.phase_name = name
        self.deadline = time.time() + (limit_s if limit_s is not None else self.step_s)

    tick = phase

    def done(self):
        self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            d = self.deadline
            if d is not None and time.time() > d:
                sys.stderr.write('bench.py watchdog: rank %d made no progress in phase %r within its limit -- exiting 17\n'
                                 % (self.rank, self.phase_name))
                sys.stderr.flush()
                os._exit(17)


class KernelTimer(object):
    """Proxy around the C-ABI library that brackets every launch of ONE kernel symbol with HIP events
    (recorded on the stream the kernel is launched on) and tallies its algorithmic FLOPs."""

    # Dominant kernel = the conv3x3 implicit GEMM with 256 px x 64 col tiles: the persistent role-split kernel of
    # csrc/igemm_pp.hip, base instantiation igemm_pp_kernel<2,2,0,false> (igemm_kernel of csrc/igemm_conv.hip for mfma='f32').
    # Per train step at batch 16 of 512^2 it serves 12 launches: the 7 forward convolutions of that shape at levels 3-4 (BatchNorm
    # partials per pixel tile) + the 5 plain data gradients.  Round 5 moved the 7 forward convolutions at levels 1-2 to their own
    # instantiation <2,2,3,false> (BatchNorm partials merged per workgroup, dc_conv3x3_stats_rows: same main loop, other symbol;
    # rounds 2-4 counted all 19 under the base symbol); the data gradients that emit BatchNorm-backward sums run <2,2,1,*>, d0a's
    # dz-on-load data gradient <2,2,0,true>.  Launches are tallied per symbol; the line reports the one with the largest total time.
    KERNELS = {
        'f16x3': ('igemm_pp_kernel<2,2,0,false>', PEAK_FP16_MFMA_TFLOPS, 3),
        'f32': ('igemm_kernel<3,3,1,1,32,4,2,2,16>', PEAK_FP32_MFMA_TFLOPS, 1),
    }
    PERWG_F16 = 'igemm_pp_kernel<2,2,3,false>'
    # entry point -> (GEMM column count is Cout / Cin, is a data gradient, index of the `stats` argument | None); every
    # one ends with N, H, W, Cin, Cout, stream.  A launch is counted only if the library's own routing query says it runs the
    # role-split kernel with > 32 columns (dc_conv3x3_pp_blocks() > 0).
    SITES = {'dc_conv3x3_fwd': ('cout', 0, 5), 'dc_conv3x3_dgrad': ('cin', 1, None), 'dc_conv3x3_fwd_f16x3': ('cout', 0, 5),
             'dc_conv3x3_dgrad_f16x3': ('cin', 1, None), 'dc_conv3x3_fwd_bnin_f16x3': ('cout', 0, 8)}
    # Round 6: the weight gradients and the joint backward kernels are timed too (the split-fp16 ones: each entry point ends with
    # N, H, W, Cin, Cout, stream).  Their symbol comes from the library's routing query; the events are recorded INSIDE the entry
    # point, around the matrix kernel only (dc_bracket_next_launch: the slab-reduction launches that follow are their own kernels
    # in a trace).  Algorithmic FLOPs: 2*9*Cin*Cout*N*H*W for a weight gradient, twice that for a joint kernel (data + weight
    # gradient); bytes (each tensor once): x + dz + dW, and da + z + x + dx + the packed weights + dW for a joint kernel.
    WGRAD_SITES = {'dc_conv3x3_wgrad_f16x3': 0, 'dc_conv3x3_wgrad_bnin_f16x3': 0, 'dc_conv3x3_wgrad_dzin_f16x3': 1,
                   'dc_conv3x3_bwd_joint_f16x3': 2}

    def __init__(self, lib):
        self._lib = lib
        self.records = {}                  # symbol -> [(e0, e1, flops, bytes)]
        self.enabled = False
        self.symbol = None

    def _wgrad_site(self, name, fn):
        kind = self.WGRAD_SITES[name]

        def wrapped(*args):
            if not self.enabled:
                return fn(*args)
            N, Hh, Ww, Cin, Cout = args[-6:-1]
            if kind == 2:
                symbol = 'bwd_joint32_kernel' if Cin == 32 else 'bwd_joint64_kernel'
                flops = 2 * 2.0 * 9 * Cin * Cout * N * Hh * Ww
                nbytes = 4.0 * (N * Hh * Ww * (2 * Cout + 2 * Cin) + 2 * 9 * Cin * Cout)
            else:
                symbol = self._lib.dc_conv3x3_wgrad_kernel_name(N, Hh, Ww, Cin, Cout, kind).decode()
                flops = 2.0 * 9 * Cin * Cout * N * Hh * Ww
                nbytes = 4.0 * (N * Hh * Ww * (Cin + (2 if kind == 1 else 1) * Cout) + 9 * Cin * Cout)
            e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
            self._lib.dc_event_create(ctypes.byref(e0))
            self._lib.dc_event_create(ctypes.byref(e1))
            self._lib.dc_bracket_next_launch(e0, e1)
            rc = fn(*args)
            self._lib.dc_bracket_next_launch(None, None)      # (an entry point that did not take the pair must not leave it to the next one)
            self.records.setdefault(symbol, []).append((e0, e1, flops, nbytes))
            return rc
        return wrapped

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self.WGRAD_SITES:
            return self._wgrad_site(name, fn)
        if name not in self.SITES:
            return fn
        colkey, dgrad, stats_at = self.SITES[name]
        f16 = name.endswith('_f16x3')

        def wrapped(*args):
            if not self.enabled:
                return fn(*args)
            N, Hh, Ww, Cin, Cout = args[-6:-1]
            ncols = Cout if colkey == 'cout' else Cin
            if not (Ww > 16 and ncols > 32):
                return fn(*args)
            symbol = None
            if f16:
                with_stats = stats_at is not None and args[stats_at] is not None
                per_wg = with_stats and args[stats_at + 1] not in (0, self._lib.dc_conv3x3_tiles(N, Hh, Ww, Cout))
                if self._lib.dc_conv3x3_pp_blocks(N, Hh, Ww, Cin, Cout, dgrad, int(with_stats)) <= 0:
                    return fn(*args)      # a 256-thread kernel serves this launch (e.g. the 32-input-channel forward layer)
                symbol = self.PERWG_F16 if per_wg else self.KERNELS['f16x3'][0]
            else:
                symbol = self.KERNELS['f32'][0]
            e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
            self._lib.dc_event_create(ctypes.byref(e0))
            self._lib.dc_event_create(ctypes.byref(e1))
            stream = args[-1]
            self._lib.dc_event_record(e0, stream)
            rc = fn(*args)
            self._lib.dc_event_record(e1, stream)
            # algorithmic bytes (input + output + weights, fp32, each moved once)
            self.records.setdefault(symbol, []).append((e0, e1, 2.0 * 9 * Cin * Cout * N * Hh * Ww, 4.0 * (N * Hh * Ww * (Cin + Cout) + 9 * Cin * Cout)))
            return rc
        return wrapped

    def summarize(self, symbol=None):
        """(launches, ms, flops) of ONE symbol: `symbol`, or -- the first time -- the one with the largest total time (kept in
        self.symbol, so that the follow-up measurement reports the same kernel)."""
        per = {}
        for sym, recs in self.records.items():
            tot_ms, tot_flops, tot_bytes, n_ok = 0.0, 0.0, 0.0, 0
            for e0, e1, flops, nbytes in recs:
                ms = ctypes.c_float()
                ok = self._lib.cdll.dc_event_elapsed_ms(e0, e1, ctypes.byref(ms)) == 0      # (never recorded: a launch the hook does not serve)
                if ok:
                    n_ok += 1
                    tot_ms += ms.value
                    tot_flops += flops
                    tot_bytes += nbytes
                self._lib.dc_event_destroy(e0)
                self._lib.dc_event_destroy(e1)
            per[sym] = (n_ok, tot_ms, tot_flops, tot_bytes)
        self.records = {}
        symbol = symbol or self.symbol or (max(per, key=lambda k: per[k][1]) if per else None)
        self.symbol = symbol
        n, tot_ms, tot_flops, tot_bytes = per.get(symbol, (0, 0.0, 0.0, 0.0))
        self.last_bytes = tot_bytes
        return n, tot_ms, tot_flops


def host_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup cpu.max quota) -- os.cpu_count() reports the
    whole 256-thread host while the GPU box's container is capped (oversubscribing oneDNN 16x runs 20x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(budget_s=25.0):
    """The oracle port (oracle/unet_torch.py: torch-CPU fp32 / oneDNN, same graph, loss and Keras-form Adam) timed
    on the host cores on a BOUNDED sample of the same workload: batch-2 train steps of 512x512 for ~budget_s."""
    import torch
    from oracle import unet_numpy as on
    from oracle.unet_torch import UNetTorch
    threads = host_cores()
    torch.set_num_threads(threads)
    try:    # keep freed activation buffers in the heap: without this glibc returns them to the OS every step and the
        libc = ctypes.CDLL('libc.so.6')     # baseline spends >half its time in page faults (2.4x slower)
        libc.mallopt(-1, 2 ** 31 - 1)       # M_TRIM_THRESHOLD
        libc.mallopt(-3, 2 ** 31 - 1)       # M_MMAP_THRESHOLD
    except Exception:
        pass
    Wt = on.init_weights(NFB)
    net = UNetTorch(Wt, NFB, dtype=torch.float32)
    state = dict(it=0, m={}, v={})
    bs = 2
    x, y = on.synthetic_batch(bs, H, W)
    masks = on.make_drop_masks(NFB, bs, H, W)
    t0 = time.time()
    net.train_step(x[:1], y[:1], state, {k: v[:1] for k, v in masks.items()})     # warm-up (oneDNN primitives)
    warm = time.time() - t0
    steps, t0 = 0, time.time()
    while True:
        net.train_step(x, y, state, masks)
        steps += 1
        dt = time.time() - t0
        if dt > budget_s or steps >= 20:
            break
    return {'value': round(steps * bs / dt, 4), 'unit': 'images/s', 'cores': int(threads), 'kind': 'port',
            'sample': '%d train steps (fwd+BCE+bwd+Keras-Adam) at batch %d of 512x512 fp32 in %.1f s (+%.1f s warm-up), '
                      'oracle/unet_torch.py on torch-CPU/oneDNN with %d threads (Keras 2.0.6/TF 1.2.1, the '
                      "reference's CPU path, is not installable offline)" % (steps, bs, dt, warm, threads)}


def kernel_source_sha():
    """Fingerprint of the sources the dominant kernel is compiled from: PMC traffic measured on another build is stale."""
    import hashlib
    h = hashlib.sha256()
    for f in ('igemm_pp.hip', 'igemm_f16x3.hip', 'igemm_common.h', 'common.h', 'wgrad_f16x3.hip', 'wgrad_common.h', 'bwd_joint.hip'):
        h.update(open(os.path.join(ROOT, 'deep_calcium_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kname):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (scripts/pmc_passes.sh ->
    profiles/pmc_traffic.json).  Counters cannot be read from inside the process, so the figure comes from the committed
    profile of THIS kernel source (sha-stamped); a stale or missing profile reports null."""
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, 'no profiles/pmc_traffic.json'
    row = (t.get('kernels') or {}).get((kname or '').replace(' ', ''))
    if row is None and t.get('kernel') == kname and 'bytes_per_launch' in t:
        row = t                              # (the one-kernel layout of rounds 1-5)
    if row is None:
        return None, 'profiles/pmc_traffic.json has no row for this kernel'
    if t.get('kernel_source_sha') != kernel_source_sha():
        return None, 'profiles/pmc_traffic.json was measured on an older build of this kernel (re-run scripts/pmc_passes.sh)'
    return row.get('bytes_per_launch'), 'rocprofv3 PMC passes of this kernel source (single stream), %s' % t.get('source', '')


def bench_infer(args, model, xd, rank, world):
    """BASELINE configs[1]: forward only (BatchNorm folded into the conv epilogues), batch 8 of 512x512, replicas only."""
    import torch
    from deep_calcium_amd import parallel
    eng, B = model.engine, xd.shape[0]
    timer = KernelTimer(eng.L)
    eng.L = timer
    for _ in range(args.warmup):
        eng.forward_infer(xd)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.forward_infer(xd)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = True
    eng.use_tapes = False                 # the instrumented forwards issue every launch from Python, through the timer proxy
    for _ in range(3):
        eng.forward_infer(xd)
    torch.cuda.synchronize()
    n_launch, k_ms, k_flops = timer.summarize()
    if rank != 0:
        return
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    _, peak, mpf = KernelTimer.KERNELS[eng.mfma]
    kname = timer.symbol
    emit_json(json.dumps({
        'metric': '512x512 summary images/sec (forward only)', 'value': round(world * B * args.steps / dt, 2),
        'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': DTYPE_LABEL[eng.mfma], 'data': 'synthetic',
        'config': {'workload': 'UNet2DS forward-only, batch=%d 512x512 random fp32, nfb=32 (BASELINE.json configs[1])' % B,
                   'parallelism': 'replicas x%d' % world},
        'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(achieved / peak, 4), 'traffic': None, 'kernel': kname, 'launches': n_launch,
                     'avg_launch_ms': round(k_ms / max(n_launch, 1), 4),
                     'algorithmic_flops_per_launch': round(k_flops / max(n_launch, 1)),
                     'mfma_flops_per_algorithmic_flop': mpf, 'matrix_pipe_frac': round(mpf * achieved / peak, 4)}}))


def bench_tta(args, model, rank):
    """BASELINE configs[4]: UNet2DSummary.predict(augmentation=True) over 19 datasets (synthetic 512x512 summary images;
    the Neurofinder data is not available offline), end to end through the reference's API: model file load, reflect
    pad, 8 augmented copies in one batch-8 forward, inverse maps, mean, threshold.  19 x 8 = 152 forwards per step."""
    import tempfile
    import torch
    from deep_calcium_amd import UNet2DSummary
    if rank != 0:
        return
    tmp = tempfile.mkdtemp(prefix='dc_tta_')
    rs = np.random.RandomState(865)
    paths = []
    for k in range(19):
        p = os.path.join(tmp, 'ds%02d.npz' % k)
        np.savez(p, series_mean=(rs.random_sample((512, 512)) * 900 + 100).astype(np.float16), name=np.array('neurofinder.%02d' % k))
        paths.append(p)
    mpath = os.path.join(tmp, 'model.hdf5')
    model.save(mpath, include_optimizer=False)
    api = UNet2DSummary(cpdir=tmp)
    for _ in range(max(1, args.warmup // 3)):
        api.predict(paths, mpath, augmentation=True)
    torch.cuda.synchronize()
    steps = max(1, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        Mp, _ = api.predict(paths, mpath, augmentation=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(steps):
        api.predict(paths, mpath, augmentation=False)
    torch.cuda.synchronize()
    dt_plain = time.perf_counter() - t1
    emit_json(json.dumps({
        'metric': '512x512 forwards/sec through UNet2DSummary.predict with 8x test-time augmentation',
        'value': round(19 * 8 * steps / dt, 2), 'unit': 'forwards/s', 'n_gpus': 1, 'steps': steps, 'warmup': max(1, args.warmup // 3),
        'ms_per_step': round(dt / steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': DTYPE_LABEL[model.engine.mfma], 'data': 'synthetic',
        'config': {'workload': 'UNet2DSummary.predict(augmentation=True), 19 synthetic 512x512 datasets x 8 TTA variants, '
                               'nfb=32, through the product API; the model file is parsed on the first call and reused while it is unchanged (BASELINE.json configs[4])',
                   'datasets_per_s': round(19 * steps / dt, 2), 'datasets_per_s_without_tta': round(19 * steps / dt_plain, 2),
                   'positive_fraction': float(np.mean([m.mean() for m in Mp]))}}))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


_JSON_FD = None


def claim_stdout():
    """From here on fd 1 of this process IS stderr: whatever a native library (RCCL, Gloo, the HIP runtime) prints to
    stdout cannot land next to the one JSON line, which goes out through a private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(line):
    data = (line + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(line + '\n')
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) from THIS process,
    which has not touched the GPU (no HIP call, only a device count), wait for them and return their exit code.
    Rank 0's stdout is piped through this process: the JSON line goes to stdout, anything else a native library prints
    there (Gloo / RCCL banners) goes to stderr with the other ranks' output -- stdout carries exactly ONE line."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()            # does not initialise the GPU on this image
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               OMP_NUM_THREADS=str(max(1, host_cores() // n)))
    if ndev < n and 'DC_DIST_BACKEND' not in env:
        if ndev < 1:
            raise SystemExit('bench.py: no GPU visible')
        # fewer GPUs than ranks (a 1-GPU box): RCCL cannot place two ranks on one device, gloo can -- a FUNCTIONAL run of
        # the N-rank path, flagged in the JSON line (config.shared_gpus)
        sys.stderr.write('bench.py: %d ranks on %d GPU(s): ranks share devices over gloo (functional run)\n' % (n, ndev))
        env['DC_DIST_BACKEND'] = 'gloo'
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    json_lines = []

    def pump(stream):
        for raw in iter(stream.readline, b''):
            line = raw.decode('utf8', 'replace')
            if line.lstrip().startswith('{"metric"'):
                json_lines.append(line.rstrip('\n'))
            else:
                sys.stderr.write(line)
    th = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
    th.start()
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:          # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    th.join(timeout=10)
    for line in json_lines:
        print(line)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None,
                    help='per-GPU batch (default: 16 for train = BASELINE configs[2], 8 for infer = configs[1])')
    ap.add_argument('--window', type=int, default=512,
                    help='square window size (default 512 = the contract line).  --window 128 --batch 20 is the training '
                         "configuration of the reference's example (examples/neurons/unet2ds_nf.py:36-40), --window 96 --batch 32 "
                         "fit()'s default (unet_2d_summary.py:333-335)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--bn', default='local', choices=['local', 'sync'],
                    help="BatchNorm under data parallelism: per-rank statistics (default) or all-reduced ('sync')")
    ap.add_argument('--mode', default='train', choices=['train', 'infer', 'tta'],
                    help="train (default, the contract line: BASELINE configs[2]/[3]); infer = forward only, batch 8 "
                         "(configs[1]); tta = UNet2DSummary.predict with 8x test-time augmentation over 19 synthetic "
                         "512x512 datasets (configs[4]) -- the last two print the same JSON shape for profiles/ and DESIGN.md")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 8 if args.mode == 'infer' else BATCH_PER_GPU

    global H, W
    H = W = int(args.window)
    if args.mode != 'train' and H != 512:
        ap.error("--window applies to --mode train only: 'infer' / 'tta' are BASELINE configs[1] / configs[4], defined at 512x512")
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))         # before anything in this process touches the GPU
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    claim_stdout()
    wd = Watchdog(int(os.environ.get('RANK', '0')))
    wd.phase('import torch', 600)                 # the first import on a fresh box pages the image in: 1-2 minutes

    import torch
    from deep_calcium_amd import parallel
    from deep_calcium_amd.model import Model, Adam

    wd.phase('process group rendezvous (%s)' % (os.environ.get('DC_DIST_BACKEND') or 'nccl'), 300)
    rank, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    shared_gpus = torch.cuda.device_count() < world
    if world > 1 and not shared_gpus and (torch.distributed.get_backend() != 'nccl'):
        # N devices are visible: the N-GPU line must be the RCCL one (gloo is the functional mode for N ranks on ONE device)
        raise SystemExit('--gpus %d with %d devices visible, but the process group is %r, not nccl (RCCL): refusing to '
                         'print a scaling line' % (world, torch.cuda.device_count(), torch.distributed.get_backend()))
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    B = args.batch
    wd.phase('model construction + first collective (parameter broadcast)', 300)
    model = Model((H, W), NFB, device=dev)
    model.compile(Adam(0.002), 'binary_crossentropy')
    eng = model.engine
    eng.bn_mode = args.bn
    parallel.broadcast_params(eng.pflat, eng.sflat)
    # synthetic shard of the global batch, resident in HBM (SURVEY 8d seeds, offset per rank)
    x, y = synthetic_batch(B, H, W, seed_x=865 + 1000 * rank, seed_y=866 + 1000 * rank)
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

    if args.mode == 'tta':
        wd.phase('tta', 900)
        return bench_tta(args, model, rank)
    if args.mode == 'infer':
        wd.phase('infer', 600)
        return bench_infer(args, model, xd, rank, world)

    timer = KernelTimer(eng.L)
    eng.L = timer
    wd.phase('barrier before warm-up', 120)
    parallel.barrier()

    for i in range(args.warmup):
        wd.tick('warm-up step %d' % i, 120 if i == 0 else None)      # the first step loads the code objects
        model.train_on_device_batch(xd, yd)
    wd.tick('drain after warm-up')
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hang_at = int(os.environ.get('DC_BENCH_TEST_HANG', '-1'))     # test hook: this rank stops at that timed step
    for i in range(args.steps):
        wd.tick('timed step %d' % i)
        if i == hang_at:
            time.sleep(1e6)
        vals = model.train_on_device_batch(xd, yd)
    wd.tick('drain after the timed steps')
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0             # this rank's own time, before it waits for the others
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    parallel.all_reduce_max(tmax)
    dt = float(tmax.item())
    per_rank_s = parallel.all_gather_floats(t_rank, dev)
    wd.phase('roofline instrumentation steps', 180)

    # ---- roofline of the dominant kernel: 2 instrumented steps outside the timed region --------------------
    timer.enabled = True
    eng.use_tapes = False                 # the instrumented steps issue every launch from Python, through the timer proxy
    model.ar_events = []                  # also bracket the part of the gradient exchange the step waits for
    eng.sync_events = [] if eng._sync_bn() else None      # ... and the per-layer 'sync' BatchNorm messages
    for _ in range(2):
        model.train_on_device_batch(xd, yd)
    torch.cuda.synchronize()
    syncbn = None
    if eng.sync_events:
        syncbn = (sum(a.elapsed_time(b) for a, b in eng.sync_events) / 2, len(eng.sync_events) // 2)
    eng.sync_events = None
    n_launch, k_ms, k_flops = timer.summarize()
    k_bytes = timer.last_bytes
    ar_exposed = [a.elapsed_time(b) for a, b in model.ar_events]
    model.ar_events = None
    ar_alone = None
    if parallel.exchange_active():        # the whole 31 MB flat gradient in ONE blocking all-reduce, nothing else running
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        parallel.all_reduce_sum(eng.gflat)
        torch.cuda.synchronize()
        parallel.barrier()
        e0.record()
        for _ in range(5):
            parallel.all_reduce_sum(eng.gflat)
        e1.record()
        torch.cuda.synchronize()
        ar_alone = e0.elapsed_time(e1) / 5
    # the same kernel without the concurrent weight-gradient stream (production overlaps them: +6.6 % step
    # throughput, but co-running kernels stretch each other's launch time)
    streams = eng.streams
    eng.streams = 1
    for _ in range(2):
        model.train_on_device_batch(xd, yd)
    torch.cuda.synchronize()
    n_iso, iso_ms, iso_flops = timer.summarize()
    eng.streams = streams
    timer.enabled = False

    if rank == 0:
        achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        _, peak, mfma_per_flop = KernelTimer.KERNELS[eng.mfma]
        kname = timer.symbol
        traffic, traffic_note = pmc_traffic(kname)
        if (H, B) != (512, 16):
            traffic, traffic_note = None, 'the committed PMC passes were taken at batch 16 of 512x512'
        out = {
            'metric': ('512x512 summary images/sec (train step)' if H == 512 else '%dx%d training windows/sec (train step)' % (H, W)),
            'value': round(world * B * args.steps / dt, 3),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE_LABEL[eng.mfma], 'data': 'synthetic',
            'config': {'workload': 'UNet2DS train step (fwd+BCE+bwd+Keras-Adam), batch=%d %dx%d per GPU, nfb=32 (%s)' % (
                                   B, H, W, 'BASELINE.json configs[2]; configs[3] at 8 GPUs' if (H, B) == (512, 16) else
                                   ('the shape of BASELINE.json configs[2] at another batch' if H == 512 else
                                    "the reference's own training window, unet_2d_summary.py:333-335 / examples/neurons/unet2ds_nf.py:36-40")),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world, 'bn': eng.bn_mode,
                       'contraction': 'fp32 operands split exactly into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 '
                                      'accumulate' if eng.mfma == 'f16x3' else 'fp32 MFMA',
                       'loss': float(vals[0]), 'shared_gpus': bool(shared_gpus),
                       'dist_backend': (torch.distributed.get_backend() if parallel.is_dist() else None)},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'kernel': kname, 'launches': n_launch,
                         'avg_launch_ms': round(k_ms / max(n_launch, 1), 4),
                         'algorithmic_flops_per_launch': round(k_flops / max(n_launch, 1)),
                         'algorithmic_bytes_per_launch': round(k_bytes / max(n_launch, 1)),
                         'traffic_over_algorithmic': round(traffic * max(n_launch, 1) / k_bytes, 3) if traffic and k_bytes else None,
                         'mfma_flops_per_algorithmic_flop': mfma_per_flop,
                         'matrix_pipe_frac': round(mfma_per_flop * achieved / peak, 4),
                         'achieved_without_concurrent_wgrad_stream': round(iso_flops / (iso_ms * 1e-3) / 1e12, 3) if iso_ms > 0 else None,
                         'streams': streams, 'traffic_note': traffic_note},
        }
        if syncbn is not None:
            # 'sync' BatchNorm: the (forward: sum, sum of squares; backward: dgamma, dbeta) messages of <= 8 KB between two
            # launches of every BatchNorm layer -- stream time from the launch in front of each all-reduce to its end
            out['syncbn_allreduce_ms'] = round(syncbn[0], 3)
            out['syncbn_allreduces_per_step'] = syncbn[1]
        if parallel.exchange_active():
            backend = torch.distributed.get_backend()
            out['rccl_ranks'] = world if backend == 'nccl' else 0      # ranks whose gradients travelled over RCCL (0: gloo run)
            out['allreduce_ms'] = round(ar_alone, 4)
            out['allreduce_exposed_ms'] = round(sum(ar_exposed) / max(len(ar_exposed), 1), 4)
            out['allreduce_bytes'] = int(eng.n_train * 4)
            out['allreduce_buckets'] = 1 if os.environ.get('DC_AR_BUCKETS', '3') == '1' else 3
            # who issues the collectives: librccl behind the C ABI (dc_comm_*: part of the backward's launch tape) or torch.distributed
            out['comm'] = 'dc_comm (librccl through the C ABI)' if parallel.native_comm(dev) is not None else 'torch.distributed'
            out['per_rank_images_per_s'] = [round(B * args.steps / t, 2) for t in per_rank_s]
        if world == 1 and not args.no_cpu_baseline and (H, B) == (512, 16):
            wd.phase('cpu_baseline', 300)
            out['cpu_baseline'] = cpu_baseline()
        emit_json(json.dumps(out))
    wd.phase('process group shutdown', 120)
    parallel.shutdown()
    wd.done()


if __name__ == '__main__':
    main()
