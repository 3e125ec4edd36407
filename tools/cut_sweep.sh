#!/bin/bash
# the step's kernels at several batch sizes and forced cuts of the tree (RNAD_BUCKET_ROWS):  tools/cut_sweep.sh "19 20 21" "0 140 281"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/cut_sweep; mkdir -p $O
for b in ${1:-19 20 21}; do
  for rows in ${2:-0 140 281}; do
    if [ "$rows" = "0" ]; then unset RNAD_BUCKET_ROWS; else export RNAD_BUCKET_ROWS=$rows; fi
    echo "=== B=2^$b rows=$rows" | tee -a $O/summary.txt
    tools/step_kernels.sh --batch-log2 $b 2>&1 | tee $O/b${b}_r${rows}.txt | tail -12 >> $O/summary.txt
  done
done
cat $O/summary.txt
