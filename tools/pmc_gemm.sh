#!/bin/bash
# SQ counters of the projection GEMM micro-benchmark (development tool; run through gpurun from the repo root).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=${1:-384}; K=${2:-128}
$R/tools/bin/gemm_bench $B $K
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_gemm_a -- $R/tools/bin/gemm_bench $B $K > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_gemm_b -- $R/tools/bin/gemm_bench $B $K > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_gemm_c -- $R/tools/bin/gemm_bench $B $K > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "abc":
    for f in glob.glob("$O/pmc_gemm_%s/*/*counter_collection.csv" % tag):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "proj_gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(tag, k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)))
PY
