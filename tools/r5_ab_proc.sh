#!/bin/bash
# per-PROCESS spread of the c3 line under environment settings: bash tools/r5_ab_proc.sh <repeats> "<env A>" "<env B>" ...
n=$1; shift
for i in $(seq 1 $n); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 100 --regions 3 --no-cpu --no-native --no-roofline --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %s %s' % ('$cfg', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min']))"
  done
done
