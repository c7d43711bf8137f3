#!/usr/bin/env python3
"""VERDICT r2 item 3 (cut the round quantum for B x H <= 4 rounds): what does a remainder of G unit groups (16 units each) cost
in the wave-pair kernel, against one full round of the one-wave kernel?  Needs the diagnostic flavour that reads
AIRCRAFT_HIP_ALL_PAIR (tools/variant_lib.sh diagenv -DAC_DIAG_ENV with UNITS=aircraft_hip); run on the GPU box:
    AIRCRAFT_HIP_LIB=$PWD/aircraft_amd/libaircraft_hip_diagenv.so python tools/pair_stub.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from tests.helpers import make_aircraft, synthetic_units
    n, tag = int(sys.argv[2]), sys.argv[3]
    ac = make_aircraft("nn", hidden=(128, 128, 128, 128), normalise=True)
    X, U = synthetic_units(n, seed=1)
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).cuda(); Ud = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).cuda()
    out = ac.step_sens(Xd, Ud, 0.01)
    for _ in range(5):
        ac.step_sens(Xd, Ud, 0.01, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        ac.step_sens(Xd, Ud, 0.01, out=out)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"case": tag, "units": n, "groups": n // 16, "kernel": ac.last_launch()[0], "ms": e0.elapsed_time(e1) / 40}))
    sys.exit(0)
def run(n, tag, all_pair):
    env = dict(os.environ)
    env["AIRCRAFT_HIP_ALL_PAIR"] = "1" if all_pair else "0"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), tag], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else r.stderr[-500:], flush=True)
run(16384, "one full round of the one-wave kernel (256 workgroups x 64 units)", False)
for g in (64, 128, 256, 512):
    run(16 * g, f"wave-pair kernel on {g} groups", True)
