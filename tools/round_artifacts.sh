#!/bin/bash
# Everything the round's measurement row is judged on, from the CURRENT build, on the GPU box:  tools/round_artifacts.sh <tag>
# Writes under gpurun_out/ (copy what should be kept into profiles/).
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
# 0. issue cost of the instruction classes on this hardware (the constants of bench.py's issue roof)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue --json $O/${tag}_valu_issue.json > $O/${tag}_valu_issue.txt 2>&1
# 3. HBM counters of the step's kernels (default mode, eager so that every launch is its own dispatch) and of K1
K="k_bucket_play_learn|k_bucket_play_count|k_bucket_learn|k_bucket_rollout|k_bucket_keys|k_bucket_scan|k_bucket_scatter|k_bucket_finish|k_mlp|k_rows_|k_row_records|k_optimizer"
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_fetch "FETCH_SIZE" "$K" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_write "WRITE_SIZE" "$K" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
# 3b. what the learner / rollout / keys kernels are bound by: SQ issue / wait counters (their own pass)
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "$K" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "k_bucket" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "k_bucket" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
tools/pmc_run.sh ${tag}_k1_fetch "FETCH_SIZE" "k_observe|vectorized_elementwise_kernel" -- python tools/k1_pmc.py > /dev/null
tools/pmc_run.sh ${tag}_k1_write "WRITE_SIZE" "k_observe|vectorized_elementwise_kernel" -- python tools/k1_pmc.py > /dev/null
head -5 $O/pmc_${tag}_k1_fetch.csv $O/pmc_${tag}_k1_write.csv
# 3d. MFMA pipe busy cycles of the MLP kernels
RNAD_NO_GRAPH=1 tools/pmc_run.sh ${tag}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "k_mlp|k_rows_forward" -- python tools/step_probe.py --steps 20 --no-graph > /dev/null
# 3c. the counter file bench.py's roofline reads (it carries the source hash of this build): written into profiles/ ON THE BOX so that the
# bench lines below already use it, and into gpurun_out/ for the way back; the static instruction mix and the class costs beside it
python tools/pmc_json.py $O/pmc_${tag}_fetch.csv $O/pmc_${tag}_write.csv $O/pmc_${tag}_sq.csv $O/${tag}_pmc.json $O/pmc_${tag}_k1_fetch.csv $O/pmc_${tag}_k1_write.csv $O/pmc_${tag}_mfma.csv $O/pmc_${tag}_tcp.csv
cp $O/${tag}_pmc.json profiles/${tag}_pmc.json
# 3e. the same passes over the configs[3] step and over the 2^22-lane step (leaf-path pair: k_bucket_play_count + k_bucket_learn_c<WEIGHTED>)
C4="--actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1"
tools/pmc_config.sh ${tag} c4 $C4 > $O/${tag}_pmc_c4.stdout 2>&1
tools/pmc_config.sh ${tag} b22 --batch-log2 22 > $O/${tag}_pmc_b22.stdout 2>&1
cp $O/${tag}_pmc_c4.json $O/${tag}_pmc_b22.json profiles/
python tools/isa_mix.py --issue $O/${tag}_valu_issue.json > $O/${tag}_isa_mix.log 2>&1
cp profiles/${tag}_isa_mix.json profiles/${tag}_valu_issue.json $O/
# 1. headline bench (default flags) + the per-GPU share of an 8-GPU run + configs[3] + fp16 observations
python bench.py > $O/${tag}_bench.log 2>&1; grep '^{' $O/${tag}_bench.log | tail -1 > $O/${tag}_bench.json.log
python bench.py --batch-log2 19 --steps 1000 --no-cpu-baseline > $O/${tag}_bench_b19.log 2>&1; grep '^{' $O/${tag}_bench_b19.log | tail -1 > $O/${tag}_bench_b19.json.log
python bench.py --batch-log2 22 --steps 500 --no-cpu-baseline > $O/${tag}_bench_b22.log 2>&1; grep '^{' $O/${tag}_bench_b22.log | tail -1 > $O/${tag}_bench_b22.json.log
python bench.py --actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1 --steps 200 --no-cpu-baseline > $O/${tag}_bench_c4.log 2>&1; grep '^{' $O/${tag}_bench_c4.log | tail -1 > $O/${tag}_bench_c4.json.log
python bench.py --obs-half --steps 500 --no-cpu-baseline > $O/${tag}_bench_half.log 2>&1; grep '^{' $O/${tag}_bench_half.log | tail -1 > $O/${tag}_bench_half.json.log
# 2. rocprofv3 kernel trace of the same bench command, and of the replayed step alone
bash tools/profile_bench.sh ${tag} --steps 300 > $O/${tag}_profile.log 2>&1
tools/step_kernels.sh > $O/${tag}_step_kernels.txt 2>&1
tools/step_kernels.sh --actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1 > $O/${tag}_step_kernels_c4.txt 2>&1
tools/step_kernels.sh --batch-log2 19 > $O/${tag}_step_kernels_b19.txt 2>&1
tools/step_kernels.sh --batch-log2 22 > $O/${tag}_step_kernels_b22.txt 2>&1
# (the MLP launches on all 2S rows: the fp32 first layer -- the default at A = 3 -- and the split-precision one forced on)
tools/step_kernels.sh --no-dedup > $O/${tag}a_step_kernels_nodedup.txt 2>&1
RNAD_MLP_SPLIT=1 tools/step_kernels.sh --no-dedup > $O/${tag}a_step_kernels_nodedup_split.txt 2>&1
python - <<PY
import json
for n in ("", "_b19", "_b22", "_c4", "_half"):
    try:
        j = json.load(open("$O/${tag}_bench%s.json.log" % n))
        r = j["roofline"]
        print(n or "default", "ms/step %.4f" % j["ms_per_step"], "value %.3e" % j["value"], "graph", j["net_evaluation"]["step_replayed_from_hipGraph"],
              "roofline", r["kernel"], r["bound"], "%.3f" % (r["frac"] or 0), "issue", (r.get("issue") or {}).get("frac"))
    except Exception as e:
        print(n, "failed", e)
PY
