#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of one csrc file, from hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed).
usage: python tools/resource_usage.py persist.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'echr_amd', 'csrc', sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
p = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics',
                    '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'], stderr=subprocess.PIPE, text=True)
cur = None
rows = []
for line in p.stderr.splitlines():
    m = re.search(r'remark: +([^:]+): (.+?) \[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ('Function Name', 'Name'):
        cur = {'name': subprocess.run(['c++filt', v], stdout=subprocess.PIPE, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print('%-78s %5s %5s %8s %6s %7s' % ('kernel', 'VGPR', 'AGPR', 'scratch', 'SGPR', 'Vspill'))
for r in rows:
    if flt in r['name']:
        print('%-78s %5s %5s %8s %6s %7s' % (r['name'][:78], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('ScratchSize [bytes/lane]', '?'),
                                             r.get('TotalSGPRs', '?'), r.get('VGPRs Spill', '?')))
