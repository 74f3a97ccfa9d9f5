#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_check.sh <tag> [notest]
# runs the GPU parity tests, a bench and a rocprofv3 kernel-trace of the bench; everything lands in gpurun_out/<tag>/
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$2" != "notest" ]; then
  timeout -k 10 400 python -m pytest tests -m gpu -q --tb=short > $out/tests.log 2>&1; echo "test_exit=$?"; tail -3 $out/tests.log
fi
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > $out/bench.json 2> $out/bench.err; echo "bench_exit=$?"; cat $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-roofline --no-native > $out/prof.log 2>&1; echo "prof_exit=$?"
