#!/bin/bash
# usage: bash tools/r5_probe.sh <out dir under gpurun_out>   -- timeline of one iteration + phase timing + bench line
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --steps 12 --warmup 3 --regions 1 --no-cpu --no-roofline --no-native --no-others > $out/trace.log 2>&1
python3 $root/tools/timeline.py $out/trace > $out/timeline.txt 2>&1; rm -rf $out/trace
cd $root
ECHR_STEP_TIMING=1 timeout -k 10 300 python3 bench.py --steps 100 --regions 3 --no-cpu --no-roofline --no-native --no-others > $out/bench_timing.json 2> $out/bench_timing.err
grep train_step $out/bench_timing.err | tail -2
python3 -c "import json;d=json.load(open('$out/bench_timing.json'));print('ms_per_step',d['ms_per_step'],d['config']['timed_regions'])"
