#!/usr/bin/env python3
"""Per-cycle diagnostics of the MHTT receding-horizon loop (eager)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np
import torch
from mhtt_track import s_bend
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
from aircraft_amd.control import MHTT, RecedingHorizon, Track
from aircraft_amd.synthetic import GLIDER, TRIM_STATE

B, N, overlap, iters = 64, int(os.environ.get("N", "50")), int(os.environ.get("OVERLAP", "30")), int(os.environ.get("ITERS", "2"))
DT = float(os.environ.get("DT", "0.01"))
dev = torch.device("cuda", 0)
ac = Aircraft(AircraftOpts(coeff_model_type="poly", coeff_model_path=os.path.join(ROOT, "tests/golden/poly_coef.npz"),
                           aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=int(os.environ.get("SUB", "1"))))
if os.environ.get("COM"):
    ac.com = [0.0131991, -1.78875e-08, 0.00313384]
track = Track(s_bend(radius=float(os.environ.get("R", "120")), sweep=float(os.environ.get("SWEEP", "0.9"))))
from aircraft_amd.control import MHTTWeights
mh = MHTT(system=ac, track=track, dt=DT, num_nodes=N, reg=float(os.environ.get("REG", "1.0")),
          weights=MHTTWeights(w_control=float(os.environ.get("WCTRL", "100"))))
rng = np.random.default_rng(0)
X0 = np.tile(np.asarray(TRIM_STATE, dtype=np.float64)[:, None], (1, B))
X0[1] += rng.uniform(-3, 3, B); X0[2] += rng.uniform(-2, 2, B); X0[3] += rng.uniform(-3, 3, B)
x0 = torch.as_tensor(X0, dtype=torch.float32, device=dev)
mh.set_progress(np.zeros(B))
loop = RecedingHorizon(mh, overlap=overlap, iterations=iters).allocate(x0, torch.zeros((N, 7, B), device=dev))
np.set_printoptions(precision=3, suppress=True, linewidth=220)
for c in range(int(os.environ.get("CYCLES", "30"))):
    xb = loop.x0.clone()
    loop.step()
    X, U = loop.X, loop.U
    pos0 = X[0, :3]
    ref, _ = mh.track_eval(mh.s0)
    V = X[:, 3:6].norm(dim=1)
    aero = ac.alpha(X[0], torch.zeros(7, B, device=dev))
    print(f"cycle {c:2d} J med {float(loop.cost.median()):10.2f} max {float(loop.cost.max()):10.2f}  s0 min/med {float(mh.s0.min()):.3f}/{float(mh.s0.median()):.3f} "
          f" V min/max {float(V.min()):.1f}/{float(V.max()):.1f}  |u|max {float(U[:, :3].abs().max()):.2f}  alpha0 max {float(aero.abs().max()) * 57.3:.1f}deg"
          f"  z {float(X[0, 2].min()):.1f}..{float(X[0, 2].max()):.1f}  nan {int((~torch.isfinite(X)).any(dim=0).any(dim=0).sum())}"
          f"  |w|max {float(X[:, 10:13].abs().max()):.2f}  dist med/max {float((loop.x0[:3] - ref).norm(dim=0).median()):.1f}/{float((loop.x0[:3] - ref).norm(dim=0).max()):.1f}")
