out=$GRAFT_REPO_ROOT/gpurun_out/pmc1
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-roofline > $out/run.log 2>&1
echo exit=$?; ls $out/*/ | head
