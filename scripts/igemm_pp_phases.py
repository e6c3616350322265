"""Per-step timing of the persistent role-split conv kernel (igemm_pp.hip, built with -DDC_IGEMM_TRACE): for sampled
workgroups, how long each role works per step and who reaches the step barrier last.
    python scripts/igemm_pp_phases.py HW Cin Cout"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'deep_calcium_amd', 'csrc')
LIB = os.path.join(ROOT, 'deep_calcium_amd', 'lib', 'libdcunet_trace.so')
os.environ['DC_LIB_PATH'] = LIB                        # before the package (and its _lib module) is imported
from deep_calcium_amd._build import SOURCES            # noqa: E402
if not os.path.exists(LIB) or any(os.path.getmtime(os.path.join(CSRC, f)) > os.path.getmtime(LIB) for f in os.listdir(CSRC)):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DDC_IGEMM_TRACE', '-o', LIB]
    for f in SOURCES:
        cmd += ['-x', 'hip', os.path.join(CSRC, f)]
    subprocess.run(cmd, check=True)
os.environ['DC_LIB_PATH'] = LIB
import torch                                            # noqa: E402
from deep_calcium_amd._lib import lib                   # noqa: E402

L = lib()
HW, Ci, Co, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16
x = torch.randn(N, HW, HW, Ci, device='cuda')
K = torch.randn(3, 3, Ci, Co, device='cuda') * 0.05
wp16 = torch.empty(L.dc_pack_weights_f16x3_floats(9, Ci, Co), device='cuda')
L.dc_pack_weights_f16x3(K.data_ptr(), wp16.data_ptr(), 9, Ci, Co, Ci * Co, Co, 1, 0, None)
z = torch.empty(N, HW, HW, Co, device='cuda')
stats = torch.zeros(L.dc_conv3x3_tiles(N, HW, HW, Co) * Co * 2, dtype=torch.float64, device='cuda')
trace = torch.zeros(8 * 3 * 512, dtype=torch.int64, device='cuda')
fn = L.cdll.dc_debug_set_pp_trace
fn.argtypes = [ctypes.c_void_p]


def run():
    L.dc_conv3x3_fwd_f16x3(x.data_ptr(), wp16.data_ptr(), None, z.data_ptr(), Co, stats.data_ptr(), 0, None, None, 0, None, 0,
                           None, 0, None, N, HW, HW, Ci, Co, None)


fn(None)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print('%d^2 %d->%d: %.1f us' % (HW, Ci, Co, e0.elapsed_time(e1) * 1e3))
assert fn(trace.data_ptr()) == 0
run()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 3, 512)
names = ['consumer A', 'consumer B', 'producer']
for cta in range(min(3, t.shape[0])):
    n = int(t[cta, 2, 0])
    if n == 0:
        continue
    P = t[cta, :, 1:n + 1].astype(np.float64)         # [role][stamp]: (done, released) pairs
    done, rel = P[:, 0::2], P[:, 1::2]
    steps = min(done.shape[1], rel.shape[1])
    work = done[:, 1:steps] - rel[:, 0:steps - 1]      # from the previous release to this step's arrival
    step = rel[0, 1:steps] - rel[0, 0:steps - 1]
    last = np.argmax(done[:, 1:steps], axis=0)
    print('workgroup sample %d: %d steps, step period mean %.0f cycles (p10 %.0f p90 %.0f)' % (cta, steps - 1, step.mean(), np.percentile(step, 10), np.percentile(step, 90)))
    for r in range(3):
        print('   %-10s work per step: mean %6.0f  p10 %6.0f  p90 %6.0f   last at the barrier in %4.1f %% of the steps'
              % (names[r], work[r].mean(), np.percentile(work[r], 10), np.percentile(work[r], 90), 100.0 * (last == r).mean()))
    nch = Ci // 16
    k = np.arange(steps - 1) + 1
    for r in (0, 1):
        mine = ((k // nch) % 2) == r
        print('   %-10s MFMA steps: %6.0f   epilogue / idle steps: %6.0f' % (names[r], work[r][mine].mean(), work[r][~mine].mean()))
