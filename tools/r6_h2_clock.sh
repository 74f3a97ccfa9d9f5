#!/bin/bash
# the clock held during the h2 product (logits shape and 4096^3), whole kernel / loads only / compute only: bash tools/r6_h2_clock.sh
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r6l; mkdir -p $out
for shape in "762 5001 1536" "4096 4096 4096"; do for mode in 0 64 128; do
  rm -rf $out/p
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p -- python3 $root/tools/h2_clock.py $mode $shape > $out/log.txt 2>&1
  c=$(find $out/p -name "*counter_collection.csv" | head -1); k=$(find $out/p -name "*kernel_trace.csv" | head -1)
  python3 - "$c" "$k" "$mode" "$shape" <<'PY'
import csv, sys
cyc = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm_h2" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[2])) if "gemm_h2" in r["Kernel_Name"]]
cyc, dur = cyc[5:], dur[5:]
n = min(len(cyc), len(dur))
if n:
    c, d = sum(cyc[:n]) / n, sum(dur[:n]) / n
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    print("shape %-16s mode %-3s  %7.1f us  %9.0f GUI-active cycles per XCD  -> %.2f GHz" % (sys.argv[4], sys.argv[3], d / 1e3, c / 8, c / 8 / d))
PY
done; done
rm -rf $out/p
