import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
from tests.test_gpu_ilqr import setup, dev
import ilqr_oracle as io
from tests.helpers import make_oracle
gpu = torch.device('cuda', 0)
ac, il, cost, X0, U = setup(gpu, 'poly', None, B=20, H=25)
Ud = dev(U, gpu)
X = il.rollout(dev(X0, gpu), Ud)
F, A, Bm, _ = il.linearise(X, Ud, want_c=False)
K, kff, dV = il.backward(X, Ud, A, Bm)
print('nan in X', torch.isnan(X).sum().item(), 'A', torch.isnan(A).sum().item(), 'K', torch.isnan(K).sum().item(), 'kff', torch.isnan(kff).sum().item())
print('max |K|', K.abs().max().item(), 'max |kff|', kff.abs().max().item())
Xc, Uc = il.forward(dev(X0, gpu), X, Ud, K, kff)
print('nan Xc per alpha', [torch.isnan(Xc[:, :, a*20:(a+1)*20]).sum().item() for a in range(3)], 'nan Uc', torch.isnan(Uc).sum().item())
f64 = lambda t: t.cpu().numpy().astype(np.float64)
Xr, Ur = io.forward(make_oracle(ac), cost, X0, f64(X), U, f64(K), f64(kff), il.alphas, 0.01)
print('numpy nan Xr', np.isnan(Xr).sum(), 'Ur', np.isnan(Ur).sum())
b = int(torch.isnan(Uc).any(dim=0).any(dim=0).nonzero()[0]) if torch.isnan(Uc).any() else -1
print('first nan col', b)
if b >= 0:
    print('Uc col', Uc[:, :, b].cpu().numpy()[:6]); print('Xc col', Xc[:6, :, b].cpu().numpy())
