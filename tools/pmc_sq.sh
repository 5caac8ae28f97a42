#!/bin/bash
# SQ counters per kernel of the default bench (run through gpurun from the repo root): vector / matrix pipe activity,
# wait breakdown and LDS conflicts, in their own passes (never combined with traces).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
W=${1:-v4_stateless_384}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq_a_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq_b_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq_c_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for tag in "abc":
    for f in glob.glob("$O/pmc_sq_%s_$W/*/*counter_collection.csv" % tag):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("__amd") or "at::" in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            for c, v in d.items():
                out[k][c] = sum(v) / len(v)
json.dump({"workload": "$W", "note": "mean per launch, summed over the device; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles (MI355X_MICROARCH.md)", "kernels": out},
          open("$O/sq_pmc_$W.json", "w"), indent=1)
for k, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k[:40].ljust(40), " ".join(f"{c[3:]}={v:.3g}" for c, v in sorted(d.items())))
    print(" " * 40, "fractions of wave time: valu-issue %.2f  wait-any %.2f  wait-inst %.2f  lds-issue %.2f" % (
        d.get("SQ_ACTIVE_INST_VALU", 0) / wc, d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_LDS", 0) / wc))
PY
