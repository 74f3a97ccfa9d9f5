#!/bin/bash
# Same-box A/B of two BUILDS of the library on the c3 iteration: tools/ab_lib.sh <alt .so under the repo> [reps]
# (build the variant with e.g.:  hipcc ... -DVARIANT -c persist.hip -o /tmp/p.o && hipcc -shared ... -o echr_amd/lib/alt/libechr_hip.so)
cd $GRAFT_REPO_ROOT; alt=$1; reps=${2:-4}
for rep in $(seq $reps); do for v in base alt; do
  if [ $v = alt ]; then export ECHR_LIB=$GRAFT_REPO_ROOT/$alt; else unset ECHR_LIB; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v c3', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done
