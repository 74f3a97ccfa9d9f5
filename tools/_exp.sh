for sl in 2 4 8; do ECHR_ATT_SLOTS=$sl timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots', $sl, d['ms_per_step'], d['roofline']['classes_ms_per_step'])"; done
