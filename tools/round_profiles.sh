#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/round_profiles.sh <round tag, e.g. r03> <commit hash>
# Regenerates everything under profiles/ that bench.py's roofline leg and DESIGN.md cite, FROM THE TREE THAT IS RUNNING:
#   <tag>_pmc_traffic.json / <tag>_pmc_mfma.json (separate --pmc passes, commit hash stored inside), <tag>_rocprof_kernel_stats_final.txt
#   (rocprofv3 --kernel-trace --stats of the bench command), <tag>_bench_final.json (the bench line, written AFTER the PMC files so that
#   it carries their traffic numbers), the same pair for --c5, and the greedy decoder's kernel stats / timings / stamps.  Results land in gpurun_out/<tag>_profiles/ (copy them into profiles/).
tag=${1:-r06}; commit=${2:-unknown}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${tag}_profiles
rm -rf $out; mkdir -p $out
cd $root
bash tools/pmc_traffic.sh ${tag}_pmc_traffic_raw > $out/pmc_traffic.log 2>&1
python3 tools/pmc_traffic_summary.py gpurun_out/${tag}_pmc_traffic_raw $commit > $out/${tag}_pmc_traffic.json && cp $out/${tag}_pmc_traffic.json profiles/
bash tools/pmc_mfma.sh ${tag}_pmc_mfma_raw > $out/pmc_mfma.log 2>&1
python3 tools/pmc_mfma_summary.py gpurun_out/${tag}_pmc_mfma_raw $commit > $out/${tag}_pmc_mfma.json && cp $out/${tag}_pmc_mfma.json profiles/
rm -rf gpurun_out/${tag}_pmc_traffic_raw gpurun_out/${tag}_pmc_mfma_raw
# the same two counter sets on BASELINE config 5 (bench.py --c5: the BIG persistent kernels, the proposal encoder's persistent kernels)
bash tools/pmc_traffic.sh ${tag}_pmc_traffic_raw5 --c5 > $out/pmc_traffic_c5.log 2>&1
python3 tools/pmc_traffic_summary.py gpurun_out/${tag}_pmc_traffic_raw5 $commit > $out/${tag}_pmc_traffic_c5.json && cp $out/${tag}_pmc_traffic_c5.json profiles/
bash tools/pmc_mfma.sh ${tag}_pmc_mfma_raw5 --c5 > $out/pmc_mfma_c5.log 2>&1
python3 tools/pmc_mfma_summary.py gpurun_out/${tag}_pmc_mfma_raw5 $commit > $out/${tag}_pmc_mfma_c5.json && cp $out/${tag}_pmc_mfma_c5.json profiles/
rm -rf gpurun_out/${tag}_pmc_traffic_raw5 gpurun_out/${tag}_pmc_mfma_raw5
# ... and on the native_f32 configuration of the headline workload (every product on v_mfma_f32_*: the environment switches are read at load)
( export ECHR_GEMM_H2=0 ECHR_GEMM_BF16X3=0 ECHR_PERSIST_H2=0
  bash tools/pmc_traffic.sh ${tag}_pmc_traffic_rawn > $out/pmc_traffic_native.log 2>&1
  python3 tools/pmc_traffic_summary.py gpurun_out/${tag}_pmc_traffic_rawn $commit > $out/${tag}_pmc_traffic_native.json && cp $out/${tag}_pmc_traffic_native.json profiles/
  bash tools/pmc_mfma.sh ${tag}_pmc_mfma_rawn > $out/pmc_mfma_native.log 2>&1
  python3 tools/pmc_mfma_summary.py gpurun_out/${tag}_pmc_mfma_rawn $commit > $out/${tag}_pmc_mfma_native.json && cp $out/${tag}_pmc_mfma_native.json profiles/
  rm -rf gpurun_out/${tag}_pmc_traffic_rawn gpurun_out/${tag}_pmc_mfma_rawn
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profn -- python3 $root/bench.py --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/profn.log 2>&1
  python3 $root/tools/prof_summary.py $out/profn 15 40 > $out/${tag}_rocprof_kernel_stats_native.txt
  python3 $root/tools/timeline.py $out/profn > $out/${tag}_timeline_native.txt 2>&1; rm -rf $out/profn )
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof.log 2>&1
python3 $root/tools/prof_summary.py $out/prof 15 40 > $out/${tag}_rocprof_kernel_stats_final.txt
python3 $root/tools/timeline.py $out/prof > $out/${tag}_timeline.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof5 -- python3 $root/bench.py --c5 --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof5.log 2>&1
python3 $root/tools/prof_summary.py $out/prof5 15 40 > $out/${tag}_rocprof_kernel_stats_c5.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profs -- python3 $root/tools/sample_bench.py 64 > $out/profs.log 2>&1
python3 $root/tools/prof_summary.py $out/profs 7 30 > $out/${tag}_rocprof_kernel_stats_sampler64.txt          # 2 warm-up + 5 timed decodes
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profs2 -- python3 $root/tools/sample_bench.py 1000 > $out/profs2.log 2>&1
python3 $root/tools/prof_summary.py $out/profs2 7 30 > $out/${tag}_rocprof_kernel_stats_sampler1000.txt
rm -rf $out/prof $out/prof5 $out/profs $out/profs2
cd $root
{ echo "# greedy decode, whole mode='eval' call (tools/sample_bench.py), commit $commit"; timeout -k 10 300 python3 tools/sample_bench.py 64 128 256 512 1000 2>/dev/null | grep N=;
  echo "# the same with the launch-per-step form (ECHR_PERSIST_SAMPLE=0)"; ECHR_PERSIST_SAMPLE=0 timeout -k 10 300 python3 tools/sample_bench.py 64 128 256 512 1000 2>/dev/null | grep N=;
  echo "# in-kernel stamps of the persistent decoder (tools/sample_stamps.py)"; timeout -k 10 200 python3 tools/sample_stamps.py 2>/dev/null | grep -v amdgpu; } > $out/${tag}_sampler.txt
{ echo "# in-kernel phase stamps of the persistent recurrences (tools/persist_stamps.py), commit $commit"; echo "## c3 forward"; timeout -k 10 200 python3 tools/persist_stamps.py 2>/dev/null | grep -v amdgpu;
  echo "## c3 reverse"; timeout -k 10 200 python3 tools/persist_stamps.py bwd 2>/dev/null | grep -v amdgpu; echo "## c5 (BIG) forward"; timeout -k 10 200 python3 tools/persist_stamps.py c5 2>/dev/null | grep -v amdgpu;
  echo "## c5 (BIG) reverse"; timeout -k 10 200 python3 tools/persist_stamps.py c5 bwd 2>/dev/null | grep -v amdgpu; } > $out/${tag}_persist_stamps.txt
timeout -k 10 400 python3 bench.py > $out/${tag}_bench_final.json 2> $out/bench.err; echo bench_exit=$?
timeout -k 10 400 python3 bench.py --c5 > $out/${tag}_bench_c5.json 2> $out/bench5.err; echo bench5_exit=$?
# BASELINE config 2 (forward + criterion only, train-mode dropout): bench line and kernel stats
timeout -k 10 300 python3 bench.py --mode fwd --no-cpu > $out/${tag}_bench_c2.json 2> $out/bench2.err; echo bench2_exit=$?
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof2 -- python3 $root/bench.py --mode fwd --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof2.log 2>&1
python3 $root/tools/prof_summary.py $out/prof2 15 30 > $out/${tag}_rocprof_kernel_stats_c2.txt; rm -rf $out/prof2
cd $root
timeout -k 10 120 python3 tools/host_time.py > $out/${tag}_host_time.txt 2>&1; timeout -k 10 120 python3 tools/host_time.py autograd >> $out/${tag}_host_time.txt 2>&1
ls -la $out
