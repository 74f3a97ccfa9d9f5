#!/bin/bash
# LDS counters of the h2 product on the logits shape (stand-alone): bash tools/r6_lds_pmc.sh
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r6l; mkdir -p $out
export H2_ONLY=logits_c
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf $out/p
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p -- python3 $root/tools/h2_bench.py > $out/log.txt 2>&1
  f=$(find $out/p -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    if "gemm_h2" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in d.items():
        print("%-42s %-34s per launch %.4g" % (k, c, v / max(1, n[(k, c)])))
PY
done
rm -rf $out/p
