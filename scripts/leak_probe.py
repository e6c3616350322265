"""Which epoch-end activity of the validation callback grows the host RSS?  Each candidate runs 12 times; prints the RSS delta per run."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deep_calcium_amd.net import UNetEngine


def rss():
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20


def measure(name, fn, n=12):
    fn(); torch.cuda.synchronize()
    r0 = rss()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print('%-52s %+8.1f MB per run' % (name, (rss() - r0) / n), flush=True)


eng = UNetEngine((128, 512), 32)
eng2 = UNetEngine((128, 512), 32)
x = torch.randn(8, 128, 512, device='cuda')
out_dev = torch.empty(114 * 65536, dtype=torch.uint8, device='cuda')
out_host = torch.empty(114 * 65536, dtype=torch.uint8).pin_memory()
L = eng.L


def events():
    evs = []
    for _ in range(15):
        e = torch.cuda.Event(); e.record(); evs.append(e)
    for e in evs: e.synchronize()


def thread_sync():
    e = torch.cuda.Event(); e.record()
    th = threading.Thread(target=e.synchronize, daemon=True); th.start(); th.join()


def thread_plain():
    th = threading.Thread(target=lambda: None, daemon=True); th.start(); th.join()


def d2h():
    for k in range(114):
        out_host[k * 65536:(k + 1) * 65536].copy_(out_dev[k * 65536:(k + 1) * 65536], non_blocking=True)


def flags():
    f = [eng._ovf[0:1].clone() for _ in range(15)]
    return torch.stack(f).cpu().numpy()


def weights():
    eng2.pflat.copy_(eng.pflat, non_blocking=True); eng2.sflat.copy_(eng.sflat, non_blocking=True)
    eng2._packed_dirty = eng2._fold_dirty = True
    eng2.forward_infer(x)


def rw():
    p = eng.forward_infer(x)
    st = torch.cuda.current_stream().cuda_stream
    for j in range(8):
        L.dc_round_window_u8(p[j].data_ptr(), 1, 128, 512, 0, 128, 0, 512, out_dev.data_ptr() + j * 65536, st)


def new_engine():
    e = UNetEngine((128, 512), 32, conv_kernel_init=None)
    e.pflat.copy_(eng.pflat)
    e.forward_infer(x)


for name, fn in (('15 events recorded + synchronised', events), ('a new thread that synchronises an event', thread_sync),
                 ('a new thread that does nothing', thread_plain), ('114 device -> pinned host copies', d2h),
                 ('15 flag clones + stack + .cpu()', flags), ('weights copied into a second engine + re-pack + forward', weights),
                 ('forward + 8 dc_round_window_u8', rw), ('a NEW engine per run (not what fit() does)', new_engine)):
    measure(name, fn)
