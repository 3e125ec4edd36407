tools/step_sweep.sh
tools/step_sweep.sh --batch-log2 17
