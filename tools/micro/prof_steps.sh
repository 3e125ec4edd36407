#!/bin/bash
# rocprofv3 kernel stats of tools/micro/tab_steps.py: LG=<log2 batch> TAB=<off|forward|full> tools/micro/prof_steps.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_steps_${LG:-20}_${TAB:-full}
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $R/tools/micro/tab_steps.py > $out.log 2>&1
grep "step:" $out.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 13
tot = 0
for r in rows:
    tot += float(r['TotalDurationNs'])/1e6/steps
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.4f} ms/step {int(r['Calls'])/steps:6.1f} calls avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:70]}")
print("sum of kernel time per step:", round(tot, 4), "ms; launches per step:", sum(int(r['Calls']) for r in rows) / steps)
PY
