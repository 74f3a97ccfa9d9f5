#!/bin/bash
# kernel stats of the native-fp32 configuration (every product on v_mfma_f32_*): bash tools/r5_native_prof.sh <out dir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export ECHR_GEMM_H2=0 ECHR_GEMM_BF16X3=0 ECHR_PERSIST_H2=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof.log 2>&1
python3 $root/tools/prof_summary.py $out/prof 15 40 > $out/native_kernel_stats.txt
python3 $root/tools/timeline.py $out/prof > $out/native_timeline.txt 2>&1; rm -rf $out/prof
head -30 $out/native_kernel_stats.txt
