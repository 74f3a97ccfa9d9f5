#!/usr/bin/env python3
"""Join ECHR_GEMM_LOG lines with the GEMM launches of a rocprofv3 kernel trace (same order); print the last iteration."""
import csv
import glob
import sys

out = sys.argv[1]
logs = [l.strip() for l in open(out + '/gemm.log') if l.startswith('[gemm]')]
rows = []
for f in glob.glob(out + '/prof/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [r for r in rows if r['Kernel_Name'].startswith(('gemm_f32_kernel', 'gemm_split_kernel', 'void echr::gemm', 'echr::gemm'))
     or 'gemm_f32_kernel' in r['Kernel_Name'] or 'gemm_split_kernel' in r['Kernel_Name']]
print('log lines', len(logs), 'gemm launches', len(g))
n = min(len(logs), len(g))
per = n // 3
tot = 0.0
agg = {}
for l, r in list(zip(logs, g))[n - per:]:
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += us
    kv = dict(t.split('=') for t in l.split()[1:] if '=' in t)
    gf = 2.0 * int(kv['M']) * int(kv['N']) * int(kv['K']) * int(kv['batch']) / 1e9
    ng = int(kv['wgs']) // max(1, ((int(kv['M']) + int(kv['tile'].split('x')[0]) - 1) // int(kv['tile'].split('x')[0])) * ((int(kv['N']) + int(kv['tile'].split('x')[1]) - 1) // int(kv['tile'].split('x')[1])) * int(kv['split']) * int(kv['batch']))
    print('%7.1f us %6.1f TF/s (x%d) %s' % (us, gf * ng / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf * ng / us / 1e-6 / 1e12 * 1e9 / 1e9, ng, l))
print('total %.1f us over %d launches' % (tot, per))
