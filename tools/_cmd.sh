python -m pytest tests -x -q -m gpu -k "persist or sample or greedy or fused_train_step or c5_full or long_events" 2>&1 | tail -3
python tools/persist_stamps.py 2>&1 | grep -v amdgpu | head -3
bash tools/ab_rounds.sh
