cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -x -k "fused or driver or full_path or c5 or flat_arena or staged" > gpurun_out/r4j/tests.log 2>&1; echo "test_exit=$?"; tail -3 gpurun_out/r4j/tests.log
bash tools/ab_env.sh ECHR_FWD_EXTRAS_LATE 1 0
