#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel calls/iteration, ms/iteration, average us."""
import csv
import glob
import sys

d, iters = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('source: %s\ntotal kernel time per iteration: %.3f ms (%d iterations)' % (f, tot / iters / 1e6, iters))
print('%-64s %9s %9s %9s %6s' % ('kernel', 'calls/it', 'ms/it', 'avg us', '%'))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    name = r['Name'].replace('echr::', '').replace('void ', '')
    print('%-64s %9.1f %9.3f %9.1f %6.1f' % (name[:64], int(r['Calls']) / iters, float(r['TotalDurationNs']) / iters / 1e6,
                                            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
