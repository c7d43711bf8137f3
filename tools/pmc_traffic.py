#!/usr/bin/env python3
"""HBM traffic of the headline step from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_round2_check.sh
(gpurun_out/r2_pmc_final/{FETCH_SIZE,WRITE_SIZE}/...): writes profiles/r02_pmc_traffic.json (read by bench.py) and a copy under
gpurun_out/.  Correction as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE x 2 (KB), WRITE_SIZE exact (KB)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "r2_pmc_final")
kb = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    by = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(base, name, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "step_sens" in r["Kernel_Name"] and r["Counter_Name"] == name:
                by["pair" if "step_sens_pair" in r["Kernel_Name"] else "main"].append(float(r["Counter_Value"]))
    if not by:
        sys.exit(f"no {name} rows under {base}")
    kb[name] = {k: sum(v[-3:]) / len(v[-3:]) for k, v in by.items()}   # the timed steps (the first rows are warm-up)
units, alg_per_unit = 204800, 1172
fetch, write = sum(kb["FETCH_SIZE"].values()), sum(kb["WRITE_SIZE"].values())
traffic = 2 * fetch * 1024 + write * 1024
prev = {}
try:
    prev = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")))
except Exception:
    pass
out = {
    "round": 2,
    "kernel": "k_nn_step_sens<8,true> (196 608 units: 12 whole rounds) + k_nn_step_sens_pair<8> (8 192 units: the remainder); per bench step = sum of both",
    "units_per_launch": units,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/gpu_round2_check.sh + tools/pmc_traffic.py, mean of the timed steps",
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "FETCH_SIZE_KB_by_kernel": kb["FETCH_SIZE"], "WRITE_SIZE_KB_by_kernel": kb["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read; confirmed on the known-byte-count calibration kernel of round 1, profiles/r01_pmc_traffic.json), WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": units * alg_per_unit,
    "ratio": traffic / (units * alg_per_unit),
    "history": prev.get("history", []) + ([{k: prev[k] for k in ("traffic_bytes_per_launch", "ratio", "kernel") if k in prev}] if prev.get("ratio") else []),
}
for path in (os.path.join(ROOT, "profiles", "r02_pmc_traffic.json"), os.path.join(ROOT, "gpurun_out", "r2_pmc_traffic.json")):
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "traffic_bytes_per_launch", "ratio")}))
