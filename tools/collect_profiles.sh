#!/bin/bash
# Copy what a tools/round_artifacts.sh <tag> run (+ step_kernels / pytest logs of the same gpurun call) left under gpurun_out/ into
# profiles/ under the round's names:  tools/collect_profiles.sh <tag> <round-prefix, e.g. r03>
set -e
tag=$1; rp=${2:-r03}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out; P=profiles
cp $O/r03_pmc.json $P/${rp}_pmc.json
for n in "" _b19 _b22 _c4 _half _profiled; do
  [ -f $O/${tag}_bench$n.json.log ] && cp $O/${tag}_bench$n.json.log $P/${rp}_bench$n.json.log
done
[ -f $O/${tag}_bench_kernel_stats.csv ] && cp $O/${tag}_bench_kernel_stats.csv $P/${rp}_bench_kernel_stats.csv
for c in fetch write sq mfma k1_fetch k1_write; do
  [ -f $O/pmc_${tag}_$c.csv ] && cp $O/pmc_${tag}_$c.csv $P/${rp}_pmc_$c.csv
done
for n in "" _c4 _b19; do
  [ -f $O/${tag}_step_kernels$n.txt ] && grep -v "rocprofv3\|simple_timer" $O/${tag}_step_kernels$n.txt > $P/${rp}_step_kernels$n.txt
done
[ -f $O/parity_errors.json ] && cp $O/parity_errors.json $P/${rp}_parity_errors.json
[ -f $O/${tag}_pytest.log ] && tail -400 $O/${tag}_pytest.log > $P/${rp}_pytest_gpu.log
ls -la $P | grep " ${rp}_"
