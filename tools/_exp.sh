for c in 1 0; do ECHR_CHAINS2=$c timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chains2', $c, d['ms_per_step'], d['config']['final_loss'])"; done
timeout -k 10 300 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -3
