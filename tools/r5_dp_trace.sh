#!/bin/bash
# kernel timeline of one data-parallel iteration (single rank on RCCL): bash tools/r5_dp_trace.sh <out dir under gpurun_out> ; env passes through
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
DP_NO_PROFILE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/tools/dp_host_profile.py > $out/trace.log 2>&1
python3 $root/tools/timeline.py $out/trace > $out/timeline.txt 2>&1; rm -rf $out/trace
grep issue $out/trace.log
