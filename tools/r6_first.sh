#!/bin/bash
# round 6, first GPU call: new tests, the native-fp32 product shapes of one iteration, a baseline bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_parity.py -q -x -k "abort or handover or step_count or rccl" > $out/tests.log 2>&1; echo "test_exit=$?"; tail -5 $out/tests.log
ECHR_GEMM_H2=0 ECHR_GEMM_BF16X3=0 ECHR_PERSIST_H2=0 ECHR_GEMM_LOG=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/native1.json 2> $out/gemm_native.log; echo "native_exit=$?"
grep "^\[gemm\]" $out/gemm_native.log | sort | uniq -c | sort -rn > $out/gemm_native_shapes.txt; cat $out/gemm_native_shapes.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu > $out/bench.json 2> $out/bench.err; echo "bench_exit=$?"; cat $out/bench.json
