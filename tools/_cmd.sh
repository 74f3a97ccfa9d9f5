python -m pytest tests -x -q -m gpu -k "fused_train_step or reference_shaped or two_rank or rehearsal" 2>&1 | tail -3
bash tools/ab_env.sh ECHR_SCRATCH_AHEAD 0 1
