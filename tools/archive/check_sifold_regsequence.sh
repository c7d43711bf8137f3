#!/bin/bash
# Guard against a defect of the ROCm 7.2 AMDGPU backend found in round 3 (profiles/r03_exp_last2_miscompile.txt):
# SIFoldOperands rewrites   %a:agpr_32 = COPY (%v:vgpr_32 = COPY %rs.subN)   with %rs a 64-bit REG_SEQUENCE as
#   %a:agpr_32 = REG_SEQUENCE %x, sub0, %y, sub1   — a 64-bit sequence into a 32-bit register: the subregister index is lost and
# both halves end up with the same value.  This compiles every translation unit of the product with the pass's output dumped
# and fails if the malformed form appears anywhere.   usage: tools/check_sifold_regsequence.sh [extra hipcc flags ...]
set -u
cd "$(dirname "$0")/.."
python3 - "$@" <<'PY'
import os, subprocess, sys, re
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
from aircraft_amd import build as b
extra = sys.argv[1:]
bad = []
def check(src):
    unit = os.path.basename(src)[:-4]
    cmd = ["hipcc", *b.CFLAGS, *b.UNIT_FLAGS.get(unit, []), *extra, "--cuda-device-only", "-S", "-mllvm", "-print-after=si-fold-operands",
           src, "-o", "/dev/null"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    hits = [l for l in p.stderr.splitlines() if re.search(r":(agpr|vgpr)_32 = REG_SEQUENCE", l)]
    return unit, p.returncode, hits
with ThreadPoolExecutor(max_workers=8) as ex:
    for unit, rc, hits in ex.map(check, b.sources()):
        print(f"{unit}: rc={rc} malformed 32-bit REG_SEQUENCE: {len(hits)}", flush=True)
        if rc != 0 or hits:
            bad.append((unit, hits[:3]))
print("RESULT:", "CLEAN" if not bad else f"AFFECTED: {bad}")
sys.exit(1 if bad else 0)
PY
