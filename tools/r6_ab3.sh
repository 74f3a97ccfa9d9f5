#!/bin/bash
# same-box A/B/C of one environment switch on the c3 iteration: bash tools/r6_ab3.sh VAR "v1 v2 v3" [reps]
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6w; mkdir -p $out
var=$1; vals=$2; reps=${3:-4}
for rep in $(seq $reps); do for v in $vals; do
  env $var=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v c3', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done | tee $out/ab3_$var.txt
