"""Host-side cost of enqueuing one train step (no device sync inside the timed region) vs its device time, with the step
replayed from launch tapes (default) and issued launch by launch from Python (DC_TAPES=0 semantics: use_tapes False).
    python scripts/host_overhead.py [batch=20] [window=128]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_calcium_amd import unet_hip, Adam
N, HW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 128)
x = torch.rand(N, HW, HW, device='cuda'); y = (torch.rand(N, HW, HW, device='cuda') < 0.1).to(torch.uint8)
for tapes in (True, False):
    m = unet_hip((HW, HW)); m.compile(Adam(0.002), loss='binary_crossentropy')
    e = m.engine
    e.use_tapes = tapes
    for _ in range(6): m.train_on_device_batch(x, y)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            e.forward_train(x, y, None); e.backward(defer_tail=True); e.adam_step(0.002)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append(((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
    print('batch %d of %dx%d, %s: host enqueue %s ms/step, total %s ms/step (%d replays)' % (
        N, HW, HW, 'tapes' if tapes else 'launch by launch', ' / '.join('%.2f' % a for a, _ in res), ' / '.join('%.2f' % b for _, b in res),
        e.tape_replays))
    # the whole train_on_device_batch (adds the metric read-back the host waits for after the forward)
    t0 = time.perf_counter()
    for _ in range(30): m.train_on_device_batch(x, y)
    torch.cuda.synchronize()
    print('   train_on_device_batch: %.2f ms/step' % ((time.perf_counter() - t0) / 30 * 1e3))
    if not tapes:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10):
            e.forward_train(x, y, None); e.backward(defer_tail=True); e.adam_step(0.002)
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats('tottime').print_stats(8)
    del m, e
