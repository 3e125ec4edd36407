#!/bin/bash
# configs[3] step kernels at forced cuts:  tools/c4_sweep.sh "0 563 450 360 288 230"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/c4_sweep; mkdir -p $O
for rows in ${1:-0 563 450 360 288 230 184}; do
  if [ "$rows" = "0" ]; then unset RNAD_BUCKET_ROWS; else export RNAD_BUCKET_ROWS=$rows; fi
  echo "=== c4 rows=$rows" | tee -a $O/summary.txt
  tools/step_kernels.sh --actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1 2>&1 | tee $O/r${rows}.txt | tail -22 >> $O/summary.txt
done
cat $O/summary.txt
