#!/bin/bash
# The counter passes of tools/round_artifacts.sh for ANOTHER configuration of the step (configs[3], 2^22 lanes, ...):
#   tools/pmc_config.sh <tag> <suffix> [step_probe args]      e.g.  tools/pmc_config.sh r06 c4 --actions 5 --transitions 4 --depth 8 --prune 7 8 --threshold 0.1
# FETCH_SIZE, WRITE_SIZE, SQ issue / wait, LDS and vector-L1 (TCP) counters, each in its own rocprofv3 run with --kernel-trace only,
# over the eagerly enqueued step (one dispatch per launch) -> gpurun_out/<tag>_pmc_<suffix>.json (tools/pmc_json.py; carries source_hash).
tag=$1; sfx=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
K="k_bucket|k_stage|k_mlp|k_rows_|k_row_records|k_optimizer|k_compact|k_leaf"
P="python tools/step_probe.py --steps 12 --no-graph $*"
export RNAD_NO_GRAPH=1
tools/pmc_run.sh ${tag}_${sfx}_fetch "FETCH_SIZE" "$K" -- $P > /dev/null
tools/pmc_run.sh ${tag}_${sfx}_write "WRITE_SIZE" "$K" -- $P > /dev/null
tools/pmc_run.sh ${tag}_${sfx}_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" "$K" -- $P > /dev/null
tools/pmc_run.sh ${tag}_${sfx}_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "$K" -- $P > /dev/null
tools/pmc_run.sh ${tag}_${sfx}_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "k_bucket|k_stage" -- $P > /dev/null
tools/pmc_run.sh ${tag}_${sfx}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "k_mlp|k_rows_forward" -- $P > /dev/null
# (no TCC_* pass: rocprofv3 aborts on that counter set on this image and then hangs in its finaliser -- it cost r06 a 30-minute call)
python tools/pmc_json.py $O/pmc_${tag}_${sfx}_fetch.csv $O/pmc_${tag}_${sfx}_write.csv $O/pmc_${tag}_${sfx}_sq.csv $O/${tag}_pmc_${sfx}.json \
       /dev/null /dev/null $O/pmc_${tag}_${sfx}_mfma.csv $O/pmc_${tag}_${sfx}_tcp.csv > $O/${tag}_pmc_${sfx}.log 2>&1
tail -3 $O/${tag}_pmc_${sfx}.log | cut -c1-400
