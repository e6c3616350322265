"""Same-box, same-process, interleaved A/B of an engine flag on the train step (two models of the same seed, alternating rounds):
    python scripts/ab_engine_flags.py stats_per_wg [window=512] [batch=16] [rounds=4] [steps=20]
Prints ms/step per round for flag = False / True (the default)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd.model import Model, Adam
flag = sys.argv[1]
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
rs = np.random.RandomState(865)
x = torch.from_numpy(rs.standard_normal((B, HW, HW)).astype(np.float32)).cuda()
y = torch.from_numpy((rs.random_sample((B, HW, HW)) < 0.126).astype(np.uint8)).cuda()
models = {}
for val in (False, True):
    m = Model((HW, HW), 32); m.compile(Adam(0.002), 'binary_crossentropy')
    assert hasattr(m.engine, flag), flag
    setattr(m.engine, flag, val)
    for _ in range(6): m.train_on_device_batch(x, y)
    models[val] = m
torch.cuda.synchronize()
res = {False: [], True: []}
for r in range(rounds):
    for val in (False, True):
        m = models[val]
        t = time.perf_counter()
        for _ in range(steps): m.train_on_device_batch(x, y)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t) / steps * 1e3)
for val in (False, True):
    print('%s=%-5s  %dx%d x %d: %s  ms/step  (mean %.3f)' % (flag, val, HW, HW, B, ' '.join('%.3f' % v for v in res[val]), sum(res[val]) / len(res[val])))
