#!/bin/bash
# Copy what a tools/round_artifacts.sh <tag> run (+ the pytest logs of the same round) left under gpurun_out/ into profiles/ (tracked):
#   tools/collect_profiles.sh <tag>
set -e
tag=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
O=gpurun_out; P=profiles
for f in pmc.json pmc_c4.json pmc_b22.json isa_mix.json valu_issue.json valu_issue.txt; do
  [ -f $O/${tag}_$f ] && cp $O/${tag}_$f $P/${tag}_$f
done
for n in "" _b19 _b22 _c4 _half _profiled; do
  [ -f $O/${tag}_bench$n.json.log ] && cp $O/${tag}_bench$n.json.log $P/${tag}_bench$n.json.log
done
[ -f $O/${tag}_bench_kernel_stats.csv ] && cp $O/${tag}_bench_kernel_stats.csv $P/${tag}_bench_kernel_stats.csv
for c in fetch write sq lds tcp mfma k1_fetch k1_write c4_fetch c4_write c4_sq c4_lds c4_tcp c4_mfma b22_fetch b22_write b22_sq b22_lds b22_tcp b22_mfma; do
  [ -f $O/pmc_${tag}_$c.csv ] && cp $O/pmc_${tag}_$c.csv $P/${tag}_pmc_$c.csv
done
for n in "" _c4 _b19 _b22; do
  [ -f $O/${tag}_step_kernels$n.txt ] && grep -v "rocprofv3\|simple_timer\|amdgpu.ids" $O/${tag}_step_kernels$n.txt > $P/${tag}_step_kernels$n.txt
done
for n in nodedup nodedup_split; do
  [ -f $O/${tag}a_step_kernels_$n.txt ] && grep -v "rocprofv3\|simple_timer\|amdgpu.ids" $O/${tag}a_step_kernels_$n.txt > $P/${tag}a_step_kernels_$n.txt
done
[ -f $O/parity_errors.json ] && cp $O/parity_errors.json $P/${tag}_parity_errors.json
[ -f $O/e2e_stats.jsonl ] && cp $O/e2e_stats.jsonl $P/${tag}_e2e_stats.jsonl
[ -f $O/${tag}_pytest_gpu.log ] && tail -400 $O/${tag}_pytest_gpu.log > $P/${tag}_pytest_gpu.log
ls $P | grep "^${tag}" | wc -l
