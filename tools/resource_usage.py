#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stderr saved to a file)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("lds", r"LDS Size \[bytes/block\]")]
for b in blocks:
    name = b.split("\n")[0].split(" ")[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ac::", "")
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\S+)", b)
        vals.append(f"{label}={m.group(1) if m else '?'}")
    print(f"{dn[:64]:64s} " + " ".join(vals))
