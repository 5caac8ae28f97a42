#!/bin/bash
# Round-2 evidence run on the GPU box (through gpurun from the repo root): kernel traces, HBM PMC passes, SQ counters and the
# bench lines of every workload; condensed afterwards with tools/summarize_profiles.py into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --steps 200 --warmup 20 > $O/r2_bench_default.json 2> $O/r2_bench_default.err
echo "default done"; tail -c 300 $O/r2_bench_default.json
rm -f $O/r2_bench_other.jsonl
for w in v4_memory_384 v4_memory_2700 v4_memory_48 train_tbptt3_384 cur_lstm144_384 cur_lstm128_384 cur_gru128_384 mlp_384 online_mlp_384 physrnn_384 physrnn_rad_384 physrnn_e3sm_384 physrnn_2700 physrnn_rad_2700 physrnn_e3sm_2700 cnn_384 cnn_train_384 cnn_train_512 cnn_train_2700; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline >> $O/r2_bench_other.jsonl 2>/dev/null
  echo "$w done"
done
for w in v4_stateless_384 v4_memory_2700 train_tbptt3_384; do STEPS=100 tools/profile.sh $w all > $O/p_$w.txt 2>&1; echo "profile $w done"; done
STEPS=100 tools/profile.sh v4_memory_384 trace > $O/p_v4_memory_384.txt 2>&1
STEPS=100 tools/profile.sh physrnn_rad_384 trace > $O/p_physrnn_rad_384.txt 2>&1
STEPS=100 tools/profile.sh physrnn_384 trace > $O/p_physrnn_384.txt 2>&1
STEPS=100 tools/profile.sh physrnn_e3sm_384 trace > $O/p_physrnn_e3sm_384.txt 2>&1
tools/pmc_sq.sh v4_memory_2700 > $O/sq_2700.txt 2>&1; echo "sq 2700 done"
tools/pmc_sq.sh v4_stateless_384 > $O/sq_384.txt 2>&1; echo "sq 384 done"
