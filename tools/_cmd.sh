cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/r4h/tests.log 2>&1; echo "test_exit=$?"; tail -4 gpurun_out/r4h/tests.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'], d['config']['final_loss'])"; done
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu --no-roofline --no-native --fused off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 autograd', d['ms_per_step'])"
timeout -k 10 200 python bench.py --c5 --steps 50 --warmup 5 --no-cpu --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'])"
timeout -k 10 300 python tools/fused_vs_autograd.py 40 2>&1 | grep -v amdgpu.ids | tail -4
