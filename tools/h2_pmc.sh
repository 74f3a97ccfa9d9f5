#!/bin/bash
# SQ counters of the h2 GEMM on two shapes (GPU box): where do the wave cycles go?
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-h2pmc}
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
export H2_ONLY=${H2_ONLY:-logits,big}
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/sq -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py > $out/sq.log 2>&1; echo sq_exit=$?
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/sq2 -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py > $out/sq2.log 2>&1; echo sq2_exit=$?
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/tcc -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py > $out/tcc.log 2>&1; echo tcc_exit=$?
