#!/usr/bin/env python3
"""HBM traffic of the headline step from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round2_check.sh
(gpurun_out/r2_pmc_final/{FETCH_SIZE,WRITE_SIZE}/...): writes profiles/r02_pmc_traffic.json (read by bench.py) and a copy under
gpurun_out/.  Correction as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE x 2 (KB), WRITE_SIZE exact (KB)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aircraft_amd.build import source_sha  # noqa: E402  (no torch, no GPU: hashes the kernel sources)
# usage: pmc_traffic.py [round [pmc dir under gpurun_out]]   (defaults: the round-3 evidence run)
RND = int(sys.argv[1]) if len(sys.argv) > 1 else 3
base = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else f"r{RND}_pmc_final")
OUT_NAME = f"r{RND:02d}_pmc_traffic.json"
kb = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    by = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(base, name, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "step_sens" in r["Kernel_Name"] and r["Counter_Name"] == name:
                kn = r["Kernel_Name"]
                by["pair" if "step_sens_pair" in kn else ("tri" if "step_sens_tri" in kn else "main")].append(float(r["Counter_Value"]))
    if not by:
        sys.exit(f"no {name} rows under {base}")
    kb[name] = {k: sum(v[-3:]) / len(v[-3:]) for k, v in by.items()}   # the timed steps (the first rows are warm-up)
units, alg_per_unit = 204800, 1172
fetch, write = sum(kb["FETCH_SIZE"].values()), sum(kb["WRITE_SIZE"].values())
traffic = 2 * fetch * 1024 + write * 1024
prev = {}
try:
    prev = json.load(open(os.path.join(ROOT, "profiles", OUT_NAME)))
except Exception:
    pass
out = {
    "round": RND,
    "source_sha": source_sha(),  # bench.py quotes this file only when the library it times was built from the same sources
    "kernel": "every k_nn_step_sens* dispatch of one ac_shoot_sens_f32 call (whole rounds + the remainder kernels); per bench step = their sum",
    "units_per_launch": units,
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/gpu_round{RND}_check.sh + tools/pmc_traffic.py, mean of the timed steps",
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "FETCH_SIZE_KB_by_kernel": kb["FETCH_SIZE"], "WRITE_SIZE_KB_by_kernel": kb["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read; confirmed on the known-byte-count calibration kernel of round 1, profiles/r01_pmc_traffic.json), WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": units * alg_per_unit,
    "ratio": traffic / (units * alg_per_unit),
    "history": prev.get("history", []) + ([{k: prev[k] for k in ("traffic_bytes_per_launch", "ratio", "kernel") if k in prev}] if prev.get("ratio") else []),
}
for path in (os.path.join(ROOT, "profiles", OUT_NAME), os.path.join(ROOT, "gpurun_out", f"r{RND}_pmc_traffic.json")):
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "traffic_bytes_per_launch", "ratio")}))
