#!/bin/bash
# generic same-box A/B of environment-switch values on the config-2 (forward + criterion) iteration: bash tools/r6_c2_ab1.sh VAR "v1 v2 ..." [reps]
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6w; mkdir -p $out
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do
  env $var=$v timeout -k 10 200 python bench.py --mode fwd --steps 50 --warmup 5 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v c2', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done | tee $out/c2ab_$var.txt
