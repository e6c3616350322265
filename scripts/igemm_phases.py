"""Where a workgroup of the implicit-GEMM kernel spends its cycles: builds libdcunet with -DDC_IGEMM_TRACE (phase
timestamps of sampled workgroups, csrc/igemm_f16x3.hip), runs one conv3x3 layer and prints the per-phase statistics.

    python scripts/igemm_phases.py HW Cin Cout [stats]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'deep_calcium_amd', 'csrc')
LIB = os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_trace.so')
os.environ['DC_LIB_PATH'] = LIB
os.environ.setdefault('DC_IGEMM_PP', '0')              # this script traces the 256-thread kernel
from deep_calcium_amd._build import SOURCES as srcs    # noqa: E402
if not os.path.exists(LIB) or any(os.path.getmtime(os.path.join(CSRC, f)) > os.path.getmtime(LIB) for f in os.listdir(CSRC)):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DDC_IGEMM_TRACE', '-o', LIB]
    for f in srcs:
        cmd += ['-x', 'hip', os.path.join(CSRC, f)]
    subprocess.run(cmd, check=True)
os.environ['DC_LIB_PATH'] = LIB
import torch                                            # noqa: E402
from deep_calcium_amd._lib import lib                   # noqa: E402

L = lib()
HW, Ci, Co, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16
with_stats = len(sys.argv) > 4
x = torch.randn(N, HW, HW, Ci, device='cuda')
K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
z = torch.empty(N, HW, HW, Co, device='cuda')
tiles = L.dc_conv3x3_tiles(N, HW, HW, Co)
stats = torch.zeros(tiles * Co * 2, dtype=torch.float64, device='cuda')
ncta = tiles * ((Co + 63) // 64)
trace = torch.zeros((ncta // 37 + 2) * 64, dtype=torch.int64, device='cuda')
fn = L.cdll.dc_debug_set_trace
fn.argtypes = [ctypes.c_void_p]


def run():
    L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr() if with_stats else None, 0, None,
                           None, 0, None, 0, None, 0, None, N, HW, HW, Ci, Co, None)


fn(None)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print('%d^2 %d->%d: %.1f us, %d workgroups, %.0f TF/s algorithmic' % (HW, Ci, Co, e0.elapsed_time(e1) * 1e3, ncta,
                                                                   2 * 9 * Ci * Co * N * HW * HW / e0.elapsed_time(e1) / 1e9))
assert fn(trace.data_ptr()) == 0
run()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 64)
t = t[t[:, 0] > 0]
n = int(t[0, 0])
ts = t[:, 1:n + 1].astype(np.float64)
d = np.diff(ts, axis=1)
nchunks = (Ci + 15) // 16
labels = ['setup+issue', 'tables/guard']
for c in range(nchunks):
    labels += ['c%d load wait' % c, 'c%d split+write' % c, 'c%d barrier1' % c, 'c%d mfma' % c, 'c%d barrier2' % c]
labels += ['epilogue']
print('%d sampled workgroups, %d stamps; cycles (mean / p10 / p90):' % (len(t), n))
for i in range(d.shape[1]):
    lab = labels[i] if i < len(labels) else 'phase %d' % i
    print('  %-16s %8.0f %8.0f %8.0f' % (lab, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
tot = ts[:, -1] - ts[:, 0]
mf = sum(d[:, i] for i in range(d.shape[1]) if i < len(labels) and labels[i].endswith('mfma'))
print('  workgroup lifetime %.0f cycles (p10 %.0f p90 %.0f); in MFMA blocks %.0f (%.0f %%); pure MFMA issue = %d'
      % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90), mf.mean(), 100 * mf.mean() / tot.mean(), nchunks * 108 * 32))
