#!/bin/bash
# Same-box A/B of two BUILDS of the library on the config-5 iteration: tools/ab_lib_c5.sh <alt .so under the repo> [reps]
cd $GRAFT_REPO_ROOT; alt=$1; reps=${2:-3}
for rep in $(seq $reps); do for v in base alt; do
  if [ $v = alt ]; then export ECHR_LIB=$GRAFT_REPO_ROOT/$alt; else unset ECHR_LIB; fi
  timeout -k 10 200 python bench.py --c5 --steps 20 --warmup 5 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v c5', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done
