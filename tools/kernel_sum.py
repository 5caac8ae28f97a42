#!/usr/bin/env python3
"""Per-call kernel time of a profiled workload: tools/kernel_sum.py <workload> <kernel name fragment that runs once per call>."""
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
n = max(int(r["Calls"]) for r in rows if sys.argv[2] in r["Name"])
tot = 0.0
for r in rows[:14]:
    per = float(r["TotalDurationNs"]) / n / 1e3
    tot += per
    print(f'{r["Name"][:60]:60s} {per:8.1f} us')
print("kernel sum per call", round(tot, 1), "us over", n, "calls")
