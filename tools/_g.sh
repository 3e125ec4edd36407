for r in 1 2 3 4; do echo rounds $r; RNAD_MLP_FWD_ROUNDS=$r bash tools/step_kernels.sh 2>&1 | grep -E "mlp_f|sum of"; done
