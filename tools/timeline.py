#!/usr/bin/env python3
"""Kernel timeline of the last iteration in a rocprofv3 kernel trace: start offset, duration, idle gap before each launch."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# iterations start with the index staging copy (several clamp_adam launches per iteration since the early range updates)
starts = [i for i, r in enumerate(rows) if 'stage_copy_kernel' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp']); prev = int(rows[a - 1]['End_Timestamp'])
busy = gap = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = (s - prev) / 1e3; d = (e - s) / 1e3
    busy += d; gap += max(g, 0)
    q = r.get('Stream_Id') or r.get('Queue_Id') or ''
    print('%9.1f  end %7.1f  dur %6.1f  gap %6.1f  q%-3s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, d, g, q, r['Kernel_Name'].replace('echr::', '').replace('void ', '')[:60]))
    prev = max(prev, e)
print('launches %d busy %.1f us gaps %.1f us span %.1f us' % (b - a, busy, gap, (prev - t0) / 1e3))
