#!/usr/bin/env python3
"""Effective shader clock under the sensitivity kernels: the -DAC_CLOCKS flavour (tools/clock_lib.sh) makes every wave report
its lifetime in shader cycles (s_memtime) and in constant 100 MHz ticks (s_memrealtime).
usage: AIRCRAFT_HIP_LIB=.../libaircraft_hip_clk.so diag_clock_ratio.py [--no-mfma] [--hidden 64,64,64] [--batch 4096]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.helpers import make_aircraft
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states
ap = argparse.ArgumentParser()
ap.add_argument("--no-mfma", action="store_true"); ap.add_argument("--hidden", default="64,64,64")
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, H = a.batch, 50
rng = np.random.default_rng(42)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
ac = make_aircraft("nn", hidden=tuple(int(h) for h in a.hidden.split(",")), use_mfma=not a.no_mfma)
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
F = torch.empty((H, 13, B), device=dev); A = torch.empty((H, 13, 13, B), device=dev); Bm = torch.empty((H, 13, 7, B), device=dev)
buf = torch.zeros(16, dtype=torch.int64, device=dev)
def reset():
    buf.zero_(); buf[4] = 2 ** 62

for _ in range(20):
    ms.linearise(X, U, out=(F, A, Bm, buf))
torch.cuda.synchronize(); reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    ms.linearise(X, U, out=(F, A, Bm, buf))
e1.record(); torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.float64)
ms_step = e0.elapsed_time(e1) / a.steps
reset(); ms.linearise(X, U, out=(F, A, Bm, buf)); torch.cuda.synchronize()
one = buf.cpu().numpy().astype(np.float64)
print(f"  one launch: mean wave lifetime {one[1] / one[2] / 100:.1f} us, longest {one[3] / 100:.1f} us, first start -> last end {(one[5] - one[4]) / 100:.1f} us; "
      f"per-XCD mean lifetime (us): {[round(v / (one[2] / 8) / 100, 1) for v in one[8:16]]}")
print(f"{ac.last_launch()[0]} B={B}: {ms_step:.4f} ms per step; waves/step {s[2] / a.steps:.0f}; wave lifetime {s[0] / s[2]:.0f} shader cycles "
      f"= {s[1] / s[2] / 100:.2f} us; effective shader clock {100 * s[0] / s[1]:.0f} MHz")
