#!/bin/bash
# kernel-time breakdown of the c3 iteration under fixed-order accumulation (ECHR_DETERMINISTIC=1): bash tools/r6_det_profile.sh
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r6d; mkdir -p $out
export ECHR_DETERMINISTIC=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py --steps 12 --warmup 3 --regions 1 --no-others --no-cpu --no-roofline --no-native > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$out/det_kernel_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[2], "w") as out:
    out.write("# bench.py --steps 12 --warmup 3 under ECHR_DETERMINISTIC=1 (15 iterations): kernel time by name, rocprofv3 --kernel-trace --stats\n")
    for r in rows[:32]:
        line = "%-72s calls %6s total %9.1f us avg %8.2f us %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot)
        print(line); out.write(line + "\n")
PY
rm -rf $out/prof
