out=$GRAFT_REPO_ROOT/gpurun_out/pmc2
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tools')
import gemm_bench as G
print(G.run('logits', 'NT', 1280, 5001, 1536, reps=10, algo=1))
print(G.run('logits', 'NT', 1280, 5001, 1536, reps=10, algo=0))
PY
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p1 -- python3 /tmp/one.py > $out/run1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p2 -- python3 /tmp/one.py > $out/run2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p3 -- python3 /tmp/one.py > $out/run3.log 2>&1
echo done; tail -2 $out/run1.log
