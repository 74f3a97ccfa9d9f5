cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
timeout -k 10 200 python tools/persist_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/stamps_c3_fwd.txt
timeout -k 10 400 python -m pytest tests -m gpu -q --tb=short -x -k "full_path or persistent_recurrence_equals or bench_layout" > gpurun_out/r4f/tests.log 2>&1; echo "test_exit=$?"; tail -3 gpurun_out/r4f/tests.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu --no-roofline --no-native 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'])"; done
