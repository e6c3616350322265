import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import unet_numpy as on
from deep_calcium_amd.net import UNetEngine, _ptr
N,H,W,nfb = [int(v) for v in sys.argv[1:5]]
eng = UNetEngine((H,W), nfb); Wt = on.init_weights(nfb, randomize_bn=True); eng.set_weights(Wt)
x,y = on.synthetic_batch(N,H,W); masks = on.make_drop_masks(nfb,N,H,W)
orc = on.UNetOracle(Wt,nfb); taps={}
loss_ref,p_ref,G_ref,_ = orc.loss_and_grads(x,y,masks,taps)
xd,yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
p = eng.forward_train(xd,yd,{k:torch.from_numpy(v).cuda() for k,v in masks.items()},update_moving=False).cpu().numpy()
A = eng._acts(N); T = eng._train_bufs(N)
print('x_d0b (A[d0a]) err', abs(A['d0a'].cpu().numpy()-taps['x_d0b']).max(), abs(taps['x_d0b']).max())
L = eng.L
# head bwd only
lo = eng.by_name['out']; pixels = N*H*W
L.dc_head_bwd(_ptr(A['d0b']), _ptr(A['p']), yd.data_ptr(), eng.pview(eng.pflat, lo, 'k'), _ptr(T['gA']), _ptr(T['part_ws']), pixels, nfb, None)
torch.cuda.synchronize()
da = T['gA'].cpu().numpy()[:pixels*nfb].reshape(N,H,W,nfb)
print('da_d0b err', abs(da-taps['da_d0b']).max(), abs(taps['da_d0b']).max())
l = eng.by_name['d0b']
blocks = L.dc_bn_bwd_blocks(pixels, nfb)
args = (_ptr(T['gA']), nfb, _ptr(T['z_d0b']), eng.stat_ptr(l,0), eng.stat_ptr(l,1), eng.pview(eng.pflat,l,'gamma'), eng.pview(eng.pflat,l,'beta'), None, 1.0, 0)
L.dc_bn_bwd_reduce(*args, _ptr(T['part_ws']), pixels, nfb, None)
L.dc_bn_bwd_finalize(_ptr(T['part_ws']), blocks, nfb, eng.pview(eng.gflat,l,'gamma'), eng.pview(eng.gflat,l,'beta'), None)
L.dc_bn_bwd_apply(*args, eng.pview(eng.gflat,l,'gamma'), eng.pview(eng.gflat,l,'beta'), _ptr(T['dz']), _ptr(T['part_ws2']), pixels, nfb, None)
torch.cuda.synchronize()
dz = T['dz'].cpu().numpy()[:pixels*nfb].reshape(N,H,W,nfb)
d = abs(dz-taps['dz_d0b']); print('dz_d0b err', d.max(), abs(taps['dz_d0b']).max(), 'n>1e-6:', (d>1e-6).sum(), np.argwhere(d>1e-5)[:10])
z = T['z_d0b'].cpu().numpy()
# numpy wgrad from HIP dz and HIP x
_, dK_np, _ = on.conv3x3_bwd(A['d0a'].cpu().numpy().astype(np.float64), Wt[0].astype(np.float64)*0+0 if False else np.zeros((3,3,nfb,nfb)), dz.astype(np.float64))
print('dK(np from hip dz,x) vs oracle', abs(dK_np-G_ref['d0b'][0]).max(), abs(G_ref['d0b'][0]).max())
