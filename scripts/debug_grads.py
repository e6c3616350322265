import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_numpy as on
from deep_calcium_amd.net import UNetEngine
N,H,W,nfb = [int(v) for v in sys.argv[1:5]]
eng = UNetEngine((H,W), nfb); Wt = on.init_weights(nfb, randomize_bn=True); eng.set_weights(Wt)
x,y = on.synthetic_batch(N,H,W); masks = on.make_drop_masks(nfb,N,H,W)
orc = on.UNetOracle(Wt,nfb)
loss_ref,p_ref,G_ref,_ = orc.loss_and_grads(x,y,masks)
xd,yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
p = eng.forward_train(xd,yd,{k:torch.from_numpy(v).cuda() for k,v in masks.items()},update_moving=False).cpu().numpy()
print('p err', abs(p-p_ref).max(), 'loss', eng.read_sums()[0]/p.size, loss_ref)
eng.backward(); G = eng.grads()
for name, ref in G_ref.items():
    print(name, ['%.2e/%.2e' % (abs(g-r.reshape(g.shape)).max(), abs(r).max()) for g,r in zip(G[name],ref)])
