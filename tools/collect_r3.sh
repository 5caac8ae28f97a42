#!/bin/bash
# Round-3 evidence run on the GPU box (through gpurun from the repo root): default bench line, kernel traces + PMC of the driver
# workloads, bench lines of the other workloads; condensed afterwards with tools/summarize_profiles.py into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --steps 200 --warmup 20 > $O/r3_bench_default.json 2> $O/r3_bench_default.err
echo "default done"
rm -f $O/r3_bench_other.jsonl
for w in v4_memory_384 v4_memory_2700 v4_memory_48 train_tbptt3_384 train_tbptt3_2700 gru128_train_tbptt3_384 gru128_train_tbptt3_2700 lstm144_train_tbptt3_384 lstm144_train_tbptt3_2700 cur_lstm144_384 cur_lstm144_2700 cur_lstm128_384 cur_gru128_384 cur_gru128_2700 physrnn_384 physrnn_rad_384 physrnn_e3sm_384 physrnn_2700 physrnn_rad_2700 physrnn_e3sm_2700 physrnn_wrapped_384 physrnn_wrapped_48 physrnn_wrapped_2700 physrnn_train_384 physrnn_train_2700 cnn_train_2700; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline >> $O/r3_bench_other.jsonl 2>/dev/null
  echo "$w done"
done
for w in v4_stateless_384 v4_memory_2700 train_tbptt3_384 train_tbptt3_2700; do STEPS=60 tools/profile.sh $w all > $O/p_$w.txt 2>&1; echo "profile $w done"; done
STEPS=100 tools/profile.sh v4_memory_384 trace > $O/p_v4_memory_384.txt 2>&1
STEPS=100 tools/profile.sh cur_lstm144_2700 trace > $O/p_cur_lstm144_2700.txt 2>&1
STEPS=50 tools/profile.sh physrnn_train_384 trace > $O/p_physrnn_train_384.txt 2>&1
STEPS=100 tools/profile.sh physrnn_wrapped_384 trace > $O/p_physrnn_wrapped_384.txt 2>&1
echo collect done
