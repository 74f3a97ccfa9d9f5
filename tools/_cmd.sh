set -e
python -m pytest tests -x -q -m gpu -k "fused_train_step" 2>&1 | tail -3
for i in 1 2 3; do
ECHR_DEFER_UPDATE=0 python bench.py --c5 --steps 200 --no-cpu --no-native --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 joined', d['ms_per_step'], d['config']['final_loss'])"
python bench.py --c5 --steps 200 --no-cpu --no-native --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 deferred', d['ms_per_step'], d['config']['final_loss'])"
done
