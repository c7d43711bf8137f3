#!/usr/bin/env python3
"""Launch the analytic sensitivity kernel (k_step_sens<default>: same 4 B/lane, 64-B-segment access pattern as the
NN kernel, no scratch, no weights) and the NN kernel on the bench workload, for PMC byte-counter calibration."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.helpers import make_aircraft
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states
dev = torch.device("cuda", 0)
B, H = 4096, 50
rng = np.random.default_rng(42)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
out = (torch.empty((H, 13, B), device=dev), torch.empty((H, 13, 13, B), device=dev), torch.empty((H, 13, 7, B), device=dev), None)
for model, hidden in (("default", None), ("nn", (128,) * 4)):
    ac = make_aircraft(model, hidden=hidden)
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    for _ in range(3):
        ms.linearise(X, U, out=out)
    torch.cuda.synchronize()
