#!/usr/bin/env python3
"""Phase breakdown of k_nn_step_sens_tiled (the "MFMA off" kernel, cfg2 net) from in-kernel s_memtime stamps.
Build first:  python aircraft_amd/build.py --diag ;  run on the GPU box.  usage: diag_stamps_tiled.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AIRCRAFT_HIP_LIB"] = os.path.join(ROOT, "aircraft_amd", "libaircraft_hip_diag.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.helpers import make_aircraft
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states
dev = torch.device("cuda", 0)
B, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 50
rng = np.random.default_rng(42)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
ac = make_aircraft("nn", hidden=(64,) * 3, use_mfma=False)
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
F = torch.empty((H, 13, B), device=dev); A = torch.empty((H, 13, 13, B), device=dev); Bm = torch.empty((H, 13, 7, B), device=dev)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
for _ in range(2):
    ms.linearise(X, U, out=(F, A, Bm, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ms.linearise(X, U, out=(F, A, Bm, stamps)); e1.record(); torch.cuda.synchronize()
name = ac.last_launch()[0]
assert name.startswith("k_nn_step_sens_tiled")
upw = 8 if name.endswith("8") else 16
s = stamps.cpu().numpy().astype(np.float64)
nw = s[12]
names = {0: "prologue (36 KB weight image -> LDS, barrier)", 1: "between forward() calls: dual rigid body + primal aero + z",
         3: "operand rows of layer 0 written", 2: "layer 0 (K = 8) + epilogue", 4: "hidden layers (2 x 16 k-steps + epilogues)",
         5: "last layer (64 -> 6) + epilogue", 6: "outputs y, J read back", 8: "last stage: dual rigid body after the network",
         9: "last stage: RK4 accumulation", 7: "final combination + normalisation, up to the stores"}
tot = s[:12].sum()
print(f"B={B}: {name} {e0.elapsed_time(e1):.3f} ms ; waves {int(nw)} ({upw} units each) ; mean cycles per wave {tot / nw:.0f}")
print(f"  ideal hidden layers: 2 x {3072 * upw // 16} v_pk_fma_f32 per stage; {27.6 * upw / 16:.1f} k per wave and step")
for i in (0, 1, 3, 2, 4, 5, 6, 8, 9, 7):
    print(f"  [{i}] {names[i]:62s} {s[i] / nw:10.0f} cyc/wave  {100 * s[i] / tot:5.1f} %")
