#!/usr/bin/env python3
"""What clock and power does the chip hold under a kernel?  Runs the sensitivity step back to back for a few seconds and
samples `rocm-smi` (sclk, power) from a helper thread.  usage: diag_clocks.py [--no-mfma] [--hidden 64,64,64] [--batch 4096]
(AIRCRAFT_HIP_LIB selects the library flavour)."""
import argparse, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.helpers import make_aircraft
from aircraft_amd.control import MultipleShooting
from aircraft_amd.synthetic import synthetic_controls, synthetic_states

ap = argparse.ArgumentParser()
ap.add_argument("--no-mfma", action="store_true"); ap.add_argument("--hidden", default="64,64,64")
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--seconds", type=float, default=4.0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, H = a.batch, 50
rng = np.random.default_rng(42)
X = torch.from_numpy(np.ascontiguousarray(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2), dtype=np.float32)).to(dev)
U = torch.from_numpy(np.ascontiguousarray(synthetic_controls(H, B, rng), dtype=np.float32)).to(dev)
ac = make_aircraft("nn", hidden=tuple(int(h) for h in a.hidden.split(",")), use_mfma=not a.no_mfma)
ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
out = (torch.empty((H, 13, B), device=dev), torch.empty((H, 13, 13, B), device=dev), torch.empty((H, 13, 7, B), device=dev), None)
ms.linearise(X, U, out=out); torch.cuda.synchronize()
samples, stop = [], False
def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            samples.append(r.stdout.strip().replace("\n", " | ")[-400:])
        except Exception as e:  # noqa: BLE001
            samples.append(repr(e))
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < a.seconds:
    for _ in range(50):
        ms.linearise(X, U, out=out)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print(f"{ac.last_launch()[0]} B={B}: {e0.elapsed_time(e1) / n:.4f} ms per step over {n} steps")
for s in samples[:: max(1, len(samples) // 6)]:
    print("  ", s)
