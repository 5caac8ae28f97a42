#!/bin/bash
# Collects the round-1 profiles on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench, and the HBM PMC counters in their OWN passes
#   (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip traces).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=${1:-v4_stateless_384}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 $R/bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_prof_$W.json 2>$O/bench_prof_$W.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_misc_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
ls $O/pmc_fetch_$W/*/ | head
cat $O/prof_$W/*/*kernel_stats.csv | cut -c1-150
