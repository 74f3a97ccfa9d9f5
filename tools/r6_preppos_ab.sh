#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6t; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_parity.py -x -q -k "joint or prepare or deferred" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for rep in 1 2 3; do for v in 0 1; do
  ECHR_PREPARE_POS=$v timeout -k 10 200 python bench.py --c5 --steps 20 --warmup 5 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('PREPARE_POS=$v c5', d['ms_per_step'], d['config']['timed_regions']['ms_per_step_min'])"
done; done | tee $out/ab.txt
