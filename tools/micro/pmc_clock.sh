cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc/mlp -- python $R/tools/mlp_bench.py 12582912 2 > $R/gpurun_out/pmc/mlp.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc/micro -- $R/tools/micro/_build/mfma_peak > $R/gpurun_out/pmc/micro.log 2>&1
ls -R $R/gpurun_out/pmc | head -30
