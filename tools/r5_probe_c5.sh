#!/bin/bash
# timeline of one config-5 iteration: bash tools/r5_probe_c5.sh <out dir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --c5 --steps 12 --warmup 3 --regions 1 --no-cpu --no-roofline --no-native --no-others > $out/trace.log 2>&1
python3 $root/tools/timeline.py $out/trace > $out/timeline.txt 2>&1; rm -rf $out/trace
