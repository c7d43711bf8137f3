import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import torch
from tests.test_gpu_ilqr import setup, dev
from aircraft_amd.control import ILQR, QuadraticCost
gpu=torch.device('cuda',0)
ac, il0, cost, X0, U = setup(gpu, "poly", None, B=48, H=40)
cost = QuadraticCost.goal((24.0, 0.0), w_goal=1.0, height=-185.0, w_height=40.0, w_lateral_speed=0.1, r=0.02, reg=1.0)
bounds = ((20.0 ** 2, 100.0 ** 2), (-np.deg2rad(10), np.deg2rad(10)), (-np.deg2rad(4), np.deg2rad(4)), (-1e30, 0.0))
for w in (0.0, 2e4, 2e5, 2e6):
    for iters in (8, 20):
        il = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.25, 0.1, 0.03), envelope_weight=w, envelope_bounds=bounds)
        X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=iters)
        rows,_ = il.envelope(X)
        a = rows[:,2].abs().amax(dim=0).cpu().numpy()
        print(w, iters, 'alpha max median', np.median(a), 'p90', np.quantile(a,.9), 'lim', np.deg2rad(4), 'cost0', float(hist[0].median()), 'costN', float(hist[-1].median()), 'umax', float(Uo.abs().max()))
