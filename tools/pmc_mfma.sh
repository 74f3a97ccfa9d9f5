#!/bin/bash
# MFMA activity per kernel over one training iteration (GPU box): SQ counters in their own rocprofv3 --pmc passes (with --kernel-trace
# only), the program directly after `--`.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that issued MFMAs
# (= 32 x N for v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md cycle-constants table); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_mfma}; shift          # further arguments go to bench.py (e.g. --c5)
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
args="--steps 3 --warmup 1 --regions 1 --no-others --no-cpu --no-roofline --no-native $*"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/busy -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/busy.log 2>&1; echo busy_exit=$?
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/mops -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/mops.log 2>&1; echo mops_exit=$?
python3 $GRAFT_REPO_ROOT/tools/pmc_mfma_summary.py $out > $out/summary.json; cat $out/summary.json | head -60
