#!/usr/bin/env python3
"""Kernel timeline of the last greedy decode in a rocprofv3 kernel trace of tools/sample_bench.py (ends with sample_finish / greedy_step)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'sample_finish' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]['Start_Timestamp']); prev = int(rows[a - 1]['End_Timestamp'])
busy = gap = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = (s - prev) / 1e3; d = (e - s) / 1e3
    busy += d; gap += max(g, 0)
    print('%9.1f  dur %7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, d, g, r['Kernel_Name'][:70]))
    prev = max(prev, e)
print('launches %d busy %.1f us gaps %.1f us span %.1f us' % (b - a, busy, gap, (prev - t0) / 1e3))
