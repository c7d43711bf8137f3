#!/usr/bin/env python3
"""Cycle time of the MHTT receding-horizon loop over the lifetime of a fresh process (eager and hipGraph alternating):
shows what is steady state and what is first-use cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np, torch
from mhtt_track import s_bend
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts
from aircraft_amd.control import MHTT, RecedingHorizon, Track
from aircraft_amd.synthetic import GLIDER, TRIM_STATE
dev = torch.device("cuda", 0)
ac = Aircraft(AircraftOpts(coeff_model_type="poly", coeff_model_path=os.path.join(ROOT, "tests", "golden", "poly_coef.npz"),
                           aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=3))
mh = MHTT(system=ac, track=Track(s_bend()), dt=0.03, num_nodes=100)
B = 1024
X0 = np.tile(np.asarray(TRIM_STATE, dtype=np.float64)[:, None], (1, B)); X0[1] += np.random.default_rng(0).uniform(-3, 3, B)
x0 = torch.as_tensor(X0, dtype=torch.float32, device=dev); U0 = torch.zeros((100, 7, B), device=dev)
t_start = time.perf_counter()
for rep in range(14):
    mh.set_progress(np.zeros(B))
    loop = RecedingHorizon(mh, overlap=60, iterations=3).allocate(x0, U0)
    if rep % 2: loop.capture()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.run(10); torch.cuda.synchronize()
    print(f"t={time.perf_counter()-t_start:6.2f}s rep {rep} {'graph' if rep%2 else 'eager'} ms/cycle {(time.perf_counter()-t0)*100:.2f}", flush=True)
