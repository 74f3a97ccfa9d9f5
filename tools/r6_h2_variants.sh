#!/bin/bash
# the h2 product's existing kernel variants on the active-row shapes of the bench iteration, stand-alone
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6i; mkdir -p $out
export H2_ONLY=logits_c,dOUT_c,g_w_logit_c,gin_x3,pall,big
for cfg in "ECHR_H2_M16=1" "ECHR_H2_M16=0" "ECHR_H2_M16=0 ECHR_H2_WN=64" "ECHR_H2_M16=0 ECHR_H2_WN=64 ECHR_H2_STAGES=4" "ECHR_H2_M16=0 ECHR_H2_STAGES=3" "ECHR_H2_M16=0 ECHR_H2_BM=256"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/h2_bench.py 2>/dev/null | cut -c1-70
done | tee $out/h2_variants.txt
