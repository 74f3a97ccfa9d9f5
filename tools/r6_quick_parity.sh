#!/bin/bash
# the tests that exercise the persistent kernels (training pairs, BIG, sampler, abort path), then an A/B of one environment switch if given
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6z; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py -x -q -k "persistent or timed_path_loss or long_events or abort or sample or greedy or c5_full or step_count" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
