#!/bin/bash
# Collects per-workload profiles on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <workload> [trace|pmc|all]
#   trace: rocprofv3 --kernel-trace --stats of `bench.py --workload W`
#   pmc:   HBM PMC counters in their OWN passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with traces)
# Results land under gpurun_out/; tools/summarize_profiles.py <workload> <tag> condenses them into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=${1:-v4_stateless_384}
MODE=${2:-all}
STEPS=${STEPS:-200}
if [ "$MODE" = trace ] || [ "$MODE" = all ]; then
  rm -rf $O/prof_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python3 $R/bench.py --workload $W --steps $STEPS --warmup 20 --no-cpu-baseline > $O/bench_prof_$W.json 2>$O/bench_prof_$W.err
  cat $O/prof_$W/*/*kernel_stats.csv | cut -c1-150
fi
if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
  rm -rf $O/pmc_fetch_$W $O/pmc_write_$W
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$W -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  ls $O/pmc_fetch_$W/*/ | head -3
fi
