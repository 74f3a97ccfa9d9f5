#!/bin/bash
# config 5 with / without the joint step (proposal encoder's backward issued from inside the caption call), alternating on one box
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6f; mkdir -p $out
for rep in 1 2; do for v in 1 0 hook; do
  ECHR_JOINT_ORDER=$([ $v = hook ] && echo hook || echo after) ECHR_JOINT_STEP=$([ $v = 0 ] && echo 0 || echo 1) timeout -k 10 200 python bench.py --c5 --steps 20 --warmup 5 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>$out/err_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('JOINT_STEP=$v', d['ms_per_step'], d['config']['timed_regions'], d['config']['final_loss'])"
done; done | tee $out/c5_ab.txt
