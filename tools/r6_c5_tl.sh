#!/bin/bash
# kernel timelines of one config-5 iteration with / without the joint step: bash tools/r6_c5_tl.sh
out=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  export ECHR_JOINT_STEP=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$v -- python3 $root/bench.py --c5 --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof$v.log 2>&1
  python3 $root/tools/timeline.py $out/prof$v > $out/timeline_joint$v.txt 2>&1; rm -rf $out/prof$v
done
