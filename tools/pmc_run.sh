#!/bin/bash
# One rocprofv3 counter pass (counters in their own run, with --kernel-trace only) over a command, summarised per kernel.
#   tools/pmc_run.sh <tag> "<COUNTER ...>" <kernel-name-regex> -- <command...>
# writes gpurun_out/pmc_<tag>/ (raw csv) and gpurun_out/pmc_<tag>.csv (per-kernel means of every counter + duration).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; counters=$2; regex=$3; shift 3
[ "$1" == "--" ] && shift
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
( cd $R && timeout -k 10 ${PMC_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out -o $tag -- "$@" ) > $out/run.log 2>&1
python $R/tools/pmc_summary.py $out "$regex" > $R/gpurun_out/pmc_$tag.csv
cat $R/gpurun_out/pmc_$tag.csv | cut -c1-220
