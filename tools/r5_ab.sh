#!/bin/bash
# same-box A/B of several environment settings: bash tools/r5_ab.sh "<env assignments A>" "<env assignments B>" ... ; 3 alternating rounds, c3 bench
for i in 1 2 3; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 150 python bench.py --steps 200 --regions 1 --no-cpu --no-native --no-roofline --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %s' % ('$cfg', d['ms_per_step']))"
  done
done
