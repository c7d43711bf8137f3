#!/usr/bin/env python3
"""Per-basic-block instruction mix of a gfx950 .s file (from hipcc -save-temps).
usage: asm_blocks.py file.s [kernel-substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = []
cur = None
kern = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = m.group(1)
    if re.match(r"^\.LBB\d+_\d+:", l) or m:
        if cur:
            blocks.append(cur)
        cur = {"kernel": kern, "name": l.split(":")[0][-24:], "line": i + 1, "mfma": 0, "scratch": 0, "acc": 0, "ds": 0,
               "valu": 0, "salu": 0, "vmem": 0, "wait": 0, "n": 0}
    elif cur is not None:
        t = l.strip()
        if not t or t[0] in ";." or t.startswith("s_endpgm"):
            continue
        cur["n"] += 1
        if t.startswith("v_mfma"): cur["mfma"] += 1
        elif t.startswith("scratch_"): cur["scratch"] += 1
        elif t.startswith("v_accvgpr"): cur["acc"] += 1
        elif t.startswith("ds_"): cur["ds"] += 1
        elif t.startswith("s_waitcnt") or t.startswith("s_nop"): cur["wait"] += 1
        elif t.startswith("global_") or t.startswith("buffer_"): cur["vmem"] += 1
        elif t.startswith("v_"): cur["valu"] += 1
        elif t.startswith("s_"): cur["salu"] += 1
if cur:
    blocks.append(cur)
sel = [b for b in blocks if want in (b["kernel"] or "")]
keys = ["mfma", "scratch", "acc", "ds", "valu", "salu", "vmem", "wait", "n"]
print("total", {k: sum(b[k] for b in sel) for k in keys}, "blocks", len(sel))
for b in sel:
    if b["mfma"] > 0 or b["scratch"] > 8 or b["n"] > 300:
        print(f"{b['name']:>24s} L{b['line']:<6d} " + " ".join(f"{k}={b[k]}" for k in keys))
