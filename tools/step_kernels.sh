#!/bin/bash
# every kernel of the default step with its average duration and launches per step (rocprofv3 --kernel-trace --stats over step_probe):
#   tools/step_kernels.sh [probe args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=/tmp/sk_$$; rm -rf $out; mkdir -p $out
N=200
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/step_probe.py --steps $N "$@" > $out/log 2>&1
tail -1 $out/log
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" $N <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1]))); N=int(sys.argv[2])
tot=0
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r['Name']); n=re.sub(r"\(.*","",n)
    c=int(r['Calls'])
    if c < N: continue
    per=c/(N+5)   # + the probe's 5 warm-up steps
    us=float(r['AverageNs'])/1e3
    tot+=us*per
    print(f"{n[:70]:70s} per_step={per:5.2f} avg_us={us:8.1f} us_per_step={us*per:8.1f}")
print("sum of kernels per step (us):", round(tot,1))
PY
