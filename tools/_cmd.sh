cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -x -k "full_path or fused or odd_shapes or flat_arena" > gpurun_out/r4m/tests.log 2>&1; echo "test_exit=$?"; tail -3 gpurun_out/r4m/tests.log
bash tools/ab_rounds.sh
