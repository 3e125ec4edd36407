#!/bin/bash
# the launches of ONE replayed step in order, with their durations and the gaps between them (rocprofv3 --kernel-trace):
#   tools/step_trace.sh [probe args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=/tmp/st_$$; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python $R/tools/step_probe.py --steps 40 "$@" > $out/log 2>&1
tail -1 $out/log
f=$(find $out -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']) for r in rows]
short=[re.sub(r"\(.*","",n)[:60] for n in names]
# the last complete step: from the last k_rows/k_mlp_forward-start pattern: find the last occurrence of the optimiser and walk back to the previous one
opt=[i for i,n in enumerate(short) if n.startswith("k_optimizer_step")]
a,b=opt[-2]+1,opt[-1]+1
prev_end=int(rows[a-1]['End_Timestamp'])
tot=0
for i in range(a,b):
    s,e=int(rows[i]['Start_Timestamp']),int(rows[i]['End_Timestamp'])
    print(f"{short[i]:62s} {(e-s)/1e3:8.1f} us   gap before {(s-prev_end)/1e3:6.1f} us   grid {rows[i].get('Grid_Size_X','?'):>8s} wg {rows[i].get('Workgroup_Size_X','?')}")
    tot+=(e-s); prev_end=e
print("kernels", round(tot/1e3,1), "us; step", round((int(rows[b-1]['End_Timestamp'])-int(rows[a-1]['End_Timestamp']))/1e3,1), "us")
PY
