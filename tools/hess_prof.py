#!/usr/bin/env python3
"""Four calls of the second-order blocks of the 4x128 surrogate at B x H = 204 800 — the program to put behind rocprofv3 --kernel-trace --stats
(round 2: k_nn_stage_tensors<8,true,0> 14.1 ms + <8,true,1> 5.9 ms + k_step_hess<2,2> 6.3 ms per call; round 3:
k_nn_stage_tensors_rev3<8> 11.2 ms + k_step_hess 6.3 ms)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import GLIDER, synthetic_controls, synthetic_states
dev = torch.device("cuda", 0)
B, H = 4096, 50
rng = np.random.default_rng(0)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
Lam = torch.randn(H, 13, B, device=dev)
out = torch.empty(H, 21, 21, B, device=dev)
ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=MlpData.synthetic((128, 128, 128, 128), seed=42),
                           aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
for _ in range(4):
    ms.hessian(X, U, Lam, out=out)
torch.cuda.synchronize()
