"""Where fit()'s per-step time goes at the reference example's configuration (128x128 windows, batch 20; synthetic datasets):
the same model driven (a) as bench.py does (resident batch, train_on_device_batch), (b) + dc_crop_augment from items each step,
(c) + the generator thread (items drawn from numpy's RNG in the background, queue of depth 1), (d) Model.fit_generator with no
callbacks, (e) with the per-batch callback protocol of fit() (History + 4 callbacks).  ms/step of 100 steps, best of 3."""
import os, sys, time, threading, queue
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_calcium_amd import UNet2DSummary, unet_hip, Adam, Callback
HW, B, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 20, 100
rs = np.random.RandomState(0)
nds = 19
S = [rs.standard_normal((512, 512)).astype(np.float32) for _ in range(nds)]
M = []
for _ in range(nds):
    m = np.zeros((512, 512))
    for _ in range(60):
        cy, cx = rs.randint(8, 504), rs.randint(8, 504)
        m[cy - 3:cy + 4, cx - 3:cx + 4] = 1
    M.append(m)
yc = [(0, 384)] * nds
u = UNet2DSummary.__new__(UNet2DSummary)
model = unet_hip((HW, HW)); model.compile(Adam(0.002), 'binary_crossentropy')
eng = model.engine
eng.set_crop_sources(S, M)
np.random.seed(1)
gen = u._device_batch_gen(S, M, ['d%d' % i for i in range(nds)], yc, B, STEPS, (HW, HW), 15)
first = next(gen)
xd, yd = (t.clone() for t in eng.crop_batch(first.items))
for _ in range(8): model.train_on_device_batch(xd, yd)
torch.cuda.synchronize()

def best(fn, reps=3):
    out = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append((time.perf_counter() - t) / STEPS * 1e3)
    return min(out), out

def a():
    for _ in range(STEPS): model.train_on_device_batch(xd, yd)
def b():
    for _ in range(STEPS): model.train_on_items(first)
def c():
    q = queue.Queue(maxsize=1); stop = threading.Event()
    def prod():
        while not stop.is_set():
            it = next(gen)
            while not stop.is_set():
                try: q.put(it, timeout=0.1); break
                except queue.Full: continue
    th = threading.Thread(target=prod, daemon=True); th.start()
    for _ in range(STEPS): model.train_on_items(q.get())
    stop.set()
    try: q.get_nowait()
    except queue.Empty: pass
    th.join()
def d():
    model.fit_generator(gen, steps_per_epoch=STEPS, epochs=1, verbose=0, callbacks=[], max_queue_size=1)
def e():
    model.fit_generator(gen, steps_per_epoch=STEPS, epochs=1, verbose=0, callbacks=[Callback() for _ in range(5)], max_queue_size=1)
def f():     # the generator alone: host time per batch of items
    for _ in range(STEPS): next(gen)
for name, fn in (('a resident batch', a), ('b + crop_augment per step', b), ('c + generator thread', c), ('d fit_generator, no callbacks', d),
                 ('e fit_generator, 5 callbacks', e), ('f item generator alone (host)', f)):
    bst, allv = best(fn)
    print('%-34s %.3f ms/step   (%s)' % (name, bst, ' '.join('%.3f' % v for v in allv)), flush=True)
