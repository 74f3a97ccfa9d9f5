#!/bin/bash
# native-fp32 iteration with / without the 128 x 128 tile, alternating on one box: bash tools/r6_native_ab.sh
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6c; mkdir -p $out
export ECHR_GEMM_H2=0 ECHR_GEMM_BF16X3=0 ECHR_PERSIST_H2=0
for rep in 1 2; do for v in 1 0; do
  ECHR_GEMM_T128=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --regions 3 --no-others --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('T128=$v', d['ms_per_step'], d['config']['timed_regions'])"
done; done | tee $out/native_ab.txt
