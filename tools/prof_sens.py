#!/usr/bin/env python3
"""One model's fused step + sensitivities at B x H units, a fixed number of launches: the program rocprofv3 wraps for
the per-model kernel traces and PMC passes (tools/gpu_r4_analytic.sh).
    prof_sens.py <poly|default|linear|real|deriv-poly|...> [B] [H] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states
from tests.helpers import make_aircraft

what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 50
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
dev = torch.device("cuda", 0)
deriv = what.startswith("deriv-")
model = what.split("-", 1)[1] if deriv else what
ac = make_aircraft("nn") if model == "real" else make_aircraft(model)
rng = np.random.default_rng(42)
Xh = synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2)
X = torch.from_numpy(np.ascontiguousarray(Xh, dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
F = torch.empty((H, 13, B), device=dev); A = torch.empty((H, 13, 13, B), device=dev); Bm = torch.empty((H, 13, 7, B), device=dev)
fn = (lambda: ms.linearise_implicit(X, U)) if deriv else (lambda: ms.linearise(X, U, out=(F, A, Bm, None)))
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    fn()
b.record(); torch.cuda.synchronize()
ms_ = a.elapsed_time(b) / iters
print(f"{what} B={B} H={H} kernel={ac.last_launch()[0]} {ms_ * 1e3:.1f} us/launch {B * H / ms_ * 1e3:.3e} steps/s {B * H * 1172 / ms_ / 1e6:.0f} GB/s", flush=True)
