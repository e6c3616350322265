import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
print(open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0])
import torch.nn.functional as F
x = torch.randn(1, 32, 512, 512); w = torch.randn(32, 32, 3, 3)
for t in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(t)
    F.conv2d(x, w, padding=1)
    t0 = time.time(); 
    for _ in range(3): F.conv2d(x, w, padding=1)
    dt = (time.time() - t0) / 3
    print('threads', t, 'conv 32->32 512^2: %.1f ms = %.0f GFLOP/s' % (dt * 1e3, 4.83 / dt))
