#!/usr/bin/env python3
"""GPU diagnostic: per-block error growth of a rollout against the float64 oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.helpers import make_aircraft, make_oracle, synthetic_problem, synthetic_units, BLOCKS, FLOORS

def blockerr(x, ref):
    out = {}
    for name, sl in BLOCKS.items():
        d = np.abs(x[..., sl, :] - ref[..., sl, :]).max(axis=-2)
        den = np.maximum(np.abs(ref[..., sl, :]).max(axis=-2), FLOORS[name])
        out[name] = d / den
    return out

model = sys.argv[1] if len(sys.argv) > 1 else "nn"
hidden = None
kw = dict(normalise=True)
if model == "lin10":
    ac = make_aircraft("linear", substeps=10, normalise=False)
    X, U = synthetic_units(777, seed=5)
    out = ac.state_update(torch.from_numpy(X).float().cuda(), torch.from_numpy(U).float().cuda(), 0.1).cpu().numpy()
    o = make_oracle(ac); ref = o.state_update(X, U, 0.1)
    e = blockerr(out, ref)
    for k, v in e.items():
        i = int(np.argmax(v)); print(k, v.max(), "unit", i)
    i = int(np.argmax(e["w"]))
    print("x", X[:, i]); print("u", U[:, i]); print("ref", ref[:, i]); print("gpu", out[:, i])
    # intermediate: substeps trace in oracle
    sys.exit(0)
ac = make_aircraft("nn", hidden=hidden, **kw)
B, H = 100, 50
X0, U = synthetic_problem(B, H, seed=17)
out = ac.rollout(torch.from_numpy(X0).float().cuda(), torch.from_numpy(U).float().cuda(), 0.01).cpu().numpy()
o = make_oracle(ac); ref = o.rollout(X0, U, 0.01)
e = blockerr(out, ref)   # each (H+1, B)
for k, v in e.items():
    kk, b = np.unravel_index(np.argmax(v), v.shape)
    print(k, "max", v.max(), "at node", kk, "inst", b, " median over inst at H:", np.median(v[-1]), " 90%:", np.quantile(v[-1], 0.9))
b = int(np.argmax(e["w"][-1]))
print("worst inst", b, "omega err vs k:", e["w"][::5, b])
print("ref omega traj", ref[::10, 10:13, b].T)
print("ref alpha-ish v", ref[::10, 3:6, b].T)
# one-step error from oracle states (no chaining): feed oracle trajectory nodes to GPU step
Xn = ref[:-1].transpose(1, 0, 2).reshape(13, -1); Un = U.transpose(1, 0, 2).reshape(7, -1)
one = ac.state_update(torch.from_numpy(Xn).float().cuda(), torch.from_numpy(Un).float().cuda(), 0.01).cpu().numpy()
ref1 = o.state_update(Xn.astype(np.float32).astype(np.float64), Un.astype(np.float32).astype(np.float64), 0.01)
e1 = blockerr(one, ref1)
for k, v in e1.items(): print("one-step", k, v.max(), np.median(v))
