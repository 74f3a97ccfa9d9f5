set -e
python -m pytest tests -x -q -m gpu -k "fused_train_step" 2>&1 | tail -3
bash tools/ab_env.sh ECHR_DEFER_UPDATE 1 split --c5
