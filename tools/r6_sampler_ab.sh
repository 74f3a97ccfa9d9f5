#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6q; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sample or sampler or greedy" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for rep in 1 2; do for v in 0 1; do echo "XCD=$v"; ECHR_PERSIST_XCD=$v timeout -k 10 300 python3 tools/sample_bench.py 64 256 1000 2>/dev/null | grep N=; done; done | tee $out/sampler_ab.txt
