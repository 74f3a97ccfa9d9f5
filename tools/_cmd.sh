python -m pytest tests -x -q -m gpu -k "persist or fused_train_step or c5_full or long_events or backward or grad" 2>&1 | tail -3
python tools/persist_stamps.py bwd 2>&1 | grep -v amdgpu | head -2
bash tools/ab_rounds.sh
