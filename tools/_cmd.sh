python -m pytest tests -x -q -m gpu -k "fused or deferred or reference_shaped or rehearsal or abandoned_prepare" 2>&1 | tail -3
bash tools/ab_env.sh ECHR_ARENA_FILL_LATE 0 1 --c5
