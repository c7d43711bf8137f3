#!/usr/bin/env python3
"""Headline kernel over batch sizes (H = 50): horizon-steps/s, workgroup rounds over the CUs, fraction of the fp32 MFMA peak.
Shows the round quantisation (one 64-unit workgroup per CU is resident) and the asymptote."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for B in (256, 512, 983, 1024, 2048, 4096, 4915, 8192, 16384, 65536):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B), "--steps", "20", "--warmup", "3",
                          "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    units = B * 50
    print(json.dumps({"batch": B, "units": units, "rounds": units / 64 / 256, "steps_per_s": d["value"],
                      "kernel_ms": d["roofline"]["kernel_ms"], "mfma_frac": d["roofline"]["frac"]}), flush=True)
