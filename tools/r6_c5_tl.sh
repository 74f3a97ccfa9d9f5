#!/bin/bash
# kernel timeline of one config-5 iteration: bash tools/r6_c5_tl.sh
out=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py --c5 --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof.log 2>&1
python3 $root/tools/timeline.py $out/prof > $out/timeline_c5.txt 2>&1; rm -rf $out/prof
