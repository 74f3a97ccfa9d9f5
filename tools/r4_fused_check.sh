#!/bin/bash
# fused-step checks on the GPU box: parity tests, host issue time of both paths, bench A/B
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}
mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -q --tb=short -x -k "fused or driver or rehearsal or raises" > $out/tests.log 2>&1; echo "test_exit=$?"; tail -5 $out/tests.log
timeout -k 10 120 python tools/host_time.py > $out/host_fused.log 2>&1; cat $out/host_fused.log | grep -v Warn
timeout -k 10 120 python tools/host_time.py autograd > $out/host_autograd.log 2>&1; cat $out/host_autograd.log | grep -v Warn
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu --no-roofline --no-native --fused on 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused on ', d['ms_per_step'], d['config']['final_loss'])"
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu --no-roofline --no-native --fused off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused off', d['ms_per_step'], d['config']['final_loss'])"
done
