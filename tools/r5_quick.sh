#!/bin/bash
# usage: bash tools/r5_quick.sh "<pytest -k expression>" "<env A>" "<env B>" ...   -- parity subset first, then the same-box A/B
set -e
k="$1"; shift
timeout -k 10 900 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_parity.py -x -q -k "$k" > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -2 gpurun_out/quick_tests.log
bash tools/r5_ab.sh "$@" | tee gpurun_out/quick_ab.log
