"""cfg5 closed loop over 1000 solves: when do instances leave the finite / in-envelope set?  (diagnostic)"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_aircraft
from aircraft_amd.control import ILQR, QuadraticCost, RecedingHorizon
from aircraft_amd.synthetic import near_trim_problem

B, H = 1024, 50
model = sys.argv[1] if len(sys.argv) > 1 else "nn"
ac = make_aircraft(model, hidden=(128, 128, 128, 128) if model == "nn" else None)
gx = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
from aircraft_amd.synthetic import cruise_problem
if gx > 0:
    cost = QuadraticCost.goal((gx, 0.5), w_goal=1.0, height=-200.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
    X0, _ = near_trim_problem(B, H, seed=11)
else:   # hold heading +x and wings level, sink freely: a regulator without a fixed goal point
    cost = QuadraticCost.cruise()
    X0 = cruise_problem(B, seed=11)
il = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))
x0 = torch.from_numpy(np.ascontiguousarray(X0, dtype=np.float32)).cuda()
U0 = torch.zeros((H, 7, B), device="cuda")
loop = RecedingHorizon(il, overlap=30, iterations=2).allocate(x0, U0).capture()
for c in range(0, 1000, 50):
    h = loop.run(50, record=True)
    fin = torch.isfinite(h).all(dim=1).all(dim=0)
    last = h[-1][:, fin]
    sp = last[3:6].norm(dim=0); w = last[10:13].norm(dim=0)
    print(f"solves {c+50:4d}: finite {int(fin.sum()):4d}  speed p50 {float(sp.median()):7.1f} max {float(sp.max()):9.1f}  |omega| p50 {float(w.median()):6.2f} max {float(w.max()):8.1f}  z p50 {float(last[2].median()):8.1f}", flush=True)
