cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tab -- python $R/tools/micro/tab_steps.py > $R/gpurun_out/prof_tab.log 2>&1
grep "tabular step" $R/gpurun_out/prof_tab.log
f=$(find $R/gpurun_out/prof_tab -name '*kernel_stats.csv' | head -1)
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 13
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {int(r['Calls'])/steps:6.1f} calls avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:80]}")
PY
