#!/usr/bin/env python3
"""Phase breakdown of k_nn_step_sens from in-kernel s_memtime stamps (diagnostic build flavor).
Build first:  python aircraft_amd/build.py --diag ;  run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AIRCRAFT_HIP_LIB"] = os.path.join(ROOT, "aircraft_amd", "libaircraft_hip_diag.so")
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
ac = make_aircraft("nn", hidden=(128,) * 4)
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
F = torch.empty((H, 13, B), device=dev); A = torch.empty((H, 13, 13, B), device=dev); Bm = torch.empty((H, 13, 7, B), device=dev)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
for _ in range(2):
    ms.linearise(X, U, out=(F, A, Bm, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ms.linearise(X, U, out=(F, A, Bm, stamps)); e1.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
nw = s[12]
names = ["prologue (weights->LDS)", "between forward() calls: rigid body (dual) + primal aero", "first layer", "acquire (DMA wait+barrier+issue)",
         "hidden layers: last-slab epilogue (rest of layer)", "last layer", "output broadcast (shuffles)", "post last stage: rigid body + RK4 combine", "stores",
         "hidden: slab 0 (value; 256 MFMA, no epilogue inside)", "hidden: slab 1 (256 MFMA + tanh epilogue of slab 0)", "hidden: slabs 2-5 (4 x 256 MFMA + scaling epilogues)"]
tot = s[:12].sum()
print(f"kernel {e0.elapsed_time(e1):.3f} ms ; waves {int(nw)} ; mean cycles per wave {tot / nw:.0f}")
print('  ideal: 256 MFMA x 32 cyc = 8192 cyc per slab; 12 hidden-layer calls per wave')
for i, n in enumerate(names):
    print(f"  [{i}] {n:58s} {s[i] / nw:10.0f} cyc/wave  {100 * s[i] / tot:5.1f} %")
