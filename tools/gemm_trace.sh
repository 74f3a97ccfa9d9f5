#!/bin/bash
# Per-shape GEMM time inside one training iteration: ECHR_GEMM_LOG lines joined (by launch order) with the rocprofv3 kernel trace.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-gemm_trace}
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
ECHR_GEMM_LOG=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-roofline > $out/run.log 2> $out/gemm.log; echo exit=$?
python3 $GRAFT_REPO_ROOT/tools/gemm_trace_join.py $out > $out/joined.txt; tail -70 $out/joined.txt
