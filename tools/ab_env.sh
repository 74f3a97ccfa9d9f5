#!/bin/bash
# same-box A/B of an environment switch: bash tools/ab_env.sh VAR A B [bench args]
var=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do
  for v in $a $b; do
    env $var=$v python bench.py --steps 300 --no-cpu --no-native --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['ms_per_step'])"
  done
done
