#!/bin/bash
# average duration of the kernels matching $2 for the library variant $1 (path to a .so, or "base"): tools/variant_time.sh <so|base> <regex> [probe args]
so=$1; rx=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=/tmp/vt_$$; rm -rf $out; mkdir -p $out
if [ "$so" != "base" ]; then export RNAD_HIP_SO=$R/$so; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/step_probe.py --steps 30 --no-graph "$@" > $out/log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" "$rx" "$so" <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r['Name']); n=re.sub(r"\(.*","",n)
    if re.search(sys.argv[2], n): print(f"{sys.argv[3]:40s} {n[:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
