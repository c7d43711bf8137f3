#!/usr/bin/env python3
"""Throughput of ac_shoot_hess_f32 at the bench size (B=4096 x H=50 units) for the analytic models."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, Quadrotor
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import GLIDER, synthetic_controls, synthetic_states

dev = torch.device("cuda", 0)
B, H = 4096, 50
rng = np.random.default_rng(0)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
Lam = torch.randn(H, 13, B, device=dev)
out = torch.empty(H, 21, 21, B, device=dev)
res = {}
from aircraft_amd import MlpData
for model in ("default", "linear", "poly", "quad", "nn3x64", "nn4x128"):
    if model == "quad":
        ac = Quadrotor()
    elif model.startswith("nn"):
        hidden = (64, 64, 64) if model == "nn3x64" else (128, 128, 128, 128)
        ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=MlpData.synthetic(hidden, seed=42),
                                   aircraft_config=AircraftConfiguration(dict(GLIDER)), physical_integration_substeps=1))
    else:
        path = {"poly": os.path.join(ROOT, "tests/golden/poly_coef.npz"), "linear": os.path.join(ROOT, "tests/golden/linearised.npz")}.get(model, "")
        if model == "linear":
            path = np.load(path)["W"] if os.path.exists(path) else np.eye(6)
        ac = Aircraft(AircraftOpts(coeff_model_type=model, coeff_model_path=path, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                                   physical_integration_substeps=1))
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    ms.hessian(X, U, Lam, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ms.hessian(X, U, Lam, out=out)
    e1.record(); torch.cuda.synchronize()
    ms_ = e0.elapsed_time(e1) / 5
    res[model] = {"ms": ms_, "units_per_s": B * H / ms_ * 1e3, "finite_frac": float(torch.isfinite(out).float().mean())}
print(json.dumps(res))
