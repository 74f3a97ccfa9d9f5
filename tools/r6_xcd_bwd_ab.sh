#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6s; mkdir -p $out
ECHR_PERSIST_XCD_BWD=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py -x -q -k "persistent_recurrence or timed_path_loss or long_events" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for rep in 1 2 3 4; do for v in 1 2; do
  ECHR_PERSIST_XCD_BWD=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('XCD_BWD=$v c3', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done | tee $out/ab.txt
