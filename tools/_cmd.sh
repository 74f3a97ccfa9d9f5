cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4n
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -x -k "fused or driver" > gpurun_out/r4n/tests.log 2>&1; echo "test_exit=$?"; tail -2 gpurun_out/r4n/tests.log
bash tools/ab_env.sh ECHR_EXTRAS_AUX 1 0
