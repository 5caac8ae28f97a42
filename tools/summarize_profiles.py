#!/usr/bin/env python3
"""Condense gpurun_out/{prof,pmc_fetch,pmc_write}_<workload> into profiles/ (tracked):
   <tag>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and <tag>_<workload>_pmc.json
   (per-kernel HBM bytes per launch: FETCH_SIZE x 1024 x 2 [gfx950 reports half of a wide coalesced
   read stream, MI355X_MICROARCH.md section HBM], WRITE_SIZE x 1024)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = sys.argv[1] if len(sys.argv) > 1 else "v4_stateless_384"
tag = sys.argv[2] if len(sys.argv) > 2 else "r1"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

def agg(pattern, counter):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)      # newest run (gpurun_out/ keeps earlier ones)
    if not f:
        return {}
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[-1])):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}

ks = sorted(glob.glob(os.path.join(G, f"prof_{w}", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(P, f"{tag}_{w}_kernel_stats.csv"))
fetch = agg(f"pmc_fetch_{w}/*/*counter_collection.csv", "FETCH_SIZE")
write = agg(f"pmc_write_{w}/*/*counter_collection.csv", "WRITE_SIZE")
out = {"workload": w, "note": "bytes per launch; fetch = FETCH_SIZE(KB)*1024*2 (gfx950 correction), write = WRITE_SIZE(KB)*1024",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("__amd"):
        continue
    fb = fetch.get(k, 0.0) * 1024 * 2
    wb = write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                         "FETCH_SIZE_raw_KB": fetch.get(k), "WRITE_SIZE_raw_KB": write.get(k)}
json.dump(out, open(os.path.join(P, f"{tag}_{w}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
