#!/usr/bin/env python3
"""Bitwise comparison of k_nn_step_sens_pair (n < half a round) with k_nn_step_sens (same units in a whole-round batch):
which outputs differ, where, by how much.  (With -ffp-contract=on: nothing.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import make_aircraft, synthetic_units
gpu = torch.device("cuda", 0)
for hidden in [(32, 32), (128,)*4]:
    ac = make_aircraft("nn", hidden=hidden, normalise=True)
    n = 4096
    X, U = synthetic_units(n, seed=77)
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(gpu); Ud = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(gpu)
    a = ac.step_sens(Xd, Ud, 0.01); print(ac.last_launch()[0])
    b = ac.step_sens(Xd.repeat(1, 4), Ud.repeat(1, 4), 0.01); print(ac.last_launch()[0])
    for name, p, q in zip(("Xn", "A", "B", "c"), a, b):
        q = q[..., :n]
        d = (p - q).abs()
        nz = (d > 0)
        rows = nz.reshape(-1, n).any(dim=1).nonzero().flatten().tolist()
        print(hidden, name, "max abs diff", float(d.max()), "frac differing", float(nz.float().mean()), "rows", rows[:20])
