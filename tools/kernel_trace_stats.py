#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace run with the warm-up dispatches left out.

`rocprofv3 --stats` averages over EVERY dispatch of a kernel, including the cold first launch and bench.py's warm-up
steps, so its average exceeds the timed region's `ms_per_step`.  This reads the per-dispatch `*_kernel_trace.csv` of the
same run, drops the first `--skip` dispatches of each kernel (bench.py: 1 naming call + W warm-up steps) and writes the
same columns as `*_kernel_stats.csv` for what is left — the dispatches of the timed region.

    python tools/kernel_trace_stats.py gpurun_out/r3_prof/<host>/<pid>_kernel_trace.csv --skip 3 [--only step_sens] > out.csv
"""
import argparse
import collections
import csv
import statistics
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=1, help="dispatches of each kernel to drop from the front")
    ap.add_argument("--only", default="", help="substring filter on the kernel name")
    a = ap.parse_args()
    dur = collections.OrderedDict()
    for r in csv.DictReader(open(a.trace)):
        name = r["Kernel_Name"]
        if a.only and a.only not in name:
            continue
        dur.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for name, d in dur.items():
        kept = d[a.skip:] if len(d) > a.skip else []
        if kept:
            rows.append((name, len(kept), sum(kept), sum(kept) / len(kept), min(kept), max(kept),
                         statistics.pstdev(kept) if len(kept) > 1 else 0.0, len(d) - len(kept)))
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "SkippedWarmupCalls"])
    for r in sorted(rows, key=lambda r: -r[2]):
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100.0 * r[2] / total, 2), r[4], r[5], round(r[6], 3), r[7]])


if __name__ == "__main__":
    main()
