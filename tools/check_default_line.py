#!/usr/bin/env python3
"""Prints the headline numbers of a default bench.py line (gpurun_out/r3_bench_default.json or a path)."""
import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
print("headline", round(d["value"]), d["unit"], round(d["ms_per_step"], 4), "ms  frac", round(d["roofline"]["frac"], 3))
print("memory_wrapper", round(d["memory_wrapper"]["ms_per_step"], 4), "train", round(d["train"]["ms_per_step"], 4), round(d["train"]["value"]))
if "shard_2700" in d:
    print("shard fwd", round(d["shard_2700"]["forward"]["ms_per_step"], 4), "train", round(d["shard_2700"]["train"]["ms_per_step"], 3))
for k, v in d.get("physrnn", {}).items():
    print(k, round(v["ms_per_step"], 4), "ms", round(v["value"]), v["unit"])
for k, v in d.get("gemm_split_optin", {}).items():
    print("gemm_split_optin", k, round(v["ms_per_step"], 4), "ms", round(v["value"]), v["unit"])
