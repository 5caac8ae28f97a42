#!/bin/bash
# HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, never combined with traces) of the HBM-bound side kernels:
# the training-data generator and the evaluation scores.  Run through gpurun from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for T in gen_bench eval_bench; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$T -- python3 $R/tools/$T.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$T -- python3 $R/tools/$T.py > /dev/null 2>&1
done
ls $O/pmc_fetch_gen_bench/*/ | head -3
