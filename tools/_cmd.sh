cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -x -k "tsrm or full_path or fused or c5 or two_rank or staged or flat_arena or accumulation or driver" > gpurun_out/r4j/tests.log 2>&1; echo "test_exit=$?"; tail -3 gpurun_out/r4j/tests.log
bash tools/ab_env.sh ECHR_ASYNC_LEVEL 2 1
