#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6p; mkdir -p $out
{ echo "## c3 forward"; timeout -k 10 200 python3 tools/persist_stamps.py 2>/dev/null | grep -v amdgpu; echo "## c3 reverse"; timeout -k 10 200 python3 tools/persist_stamps.py bwd 2>/dev/null | grep -v amdgpu; } > $out/stamps_xcd.txt; cat $out/stamps_xcd.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/full_gpu_tests.log 2>&1; echo test_exit=$?; tail -4 $out/full_gpu_tests.log
