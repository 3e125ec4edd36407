#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box): tools/profile_bench.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
grep "^{" $out/bench.log | tail -1 > $R/gpurun_out/${tag}_bench_profiled.json.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${tag}_bench_kernel_stats.csv
head -12 $R/gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-200
