#!/usr/bin/env python3
"""GPU diagnostic: worst unit of a chained-substep state_update against the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.helpers import make_aircraft, make_oracle, synthetic_units, f32_exact, well_conditioned, BLOCKS, FLOORS
model = sys.argv[1]; ns = int(sys.argv[2]); dt = float(sys.argv[3])
ac = make_aircraft(model, substeps=ns, normalise=False)
X, U = synthetic_units(777, seed=5); X = f32_exact(X); U = f32_exact(U)
out = ac.state_update(torch.from_numpy(X).float().cuda(), torch.from_numpy(U).float().cuda(), dt).cpu().numpy()
o = make_oracle(ac)
ok, ref = well_conditioned(o, X, U, dt, tol=5e-7, rollout=False)
for name, sl in BLOCKS.items():
    d = np.abs(out[sl] - ref[sl]).max(axis=0); den = np.maximum(np.abs(ref[sl]).max(axis=0), FLOORS[name])
    r = np.where(ok, d / den, 0)
    i = int(np.argmax(r)); print(name, "max", r.max(), "unit", i, "median", np.median(r[ok]))
i = int(np.argmax(np.where(ok, np.abs(out[10:13] - ref[10:13]).max(axis=0) / np.maximum(np.abs(ref[10:13]).max(axis=0), 0.1), 0)))
np.set_printoptions(precision=9, linewidth=200)
print("x0 ", X[:, i]); print("u  ", U[:, i]); print("ref", ref[:, i]); print("gpu", out[:, i].astype(np.float64)); print("err", out[:, i] - ref[:, i])
# substep trace with the oracle using fp32-rounded state each substep (emulates pure fp32 carry)
o1 = make_oracle(make_aircraft(model, substeps=1, normalise=False))
x = X[:, i:i+1].copy(); xs = x.copy()
for s in range(ns):
    x = o1.state_update(x, U[:, i:i+1], dt / ns)
    a = o1.aero(x, U[:, i:i+1])
    print("substep", s, "omega", x[10:13, 0], "alpha", a[4, 0], "beta", a[5, 0], "V", a[3, 0])
