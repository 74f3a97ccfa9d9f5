#!/usr/bin/env python3
"""Per-kernel MFMA activity from the two rocprofv3 --pmc passes of tools/pmc_mfma.sh.  Output JSON: kernel -> launches, average
duration, MFMA-busy cycles per launch, and mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the share of
all SIMD-cycles of the chip (256 CUs x 4) during the dispatch in which an MFMA was executing."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]


def name(k):
    k = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('echr::', '')
    k = k.split('<')[0] if k.startswith(('gemm_f32_kernel', 'gemm_f32_t128_kernel', 'gemm_h2_kernel', 'gemm_h2m16_kernel', 'dec_persist')) else k
    k = 'gemm_f32_kernel' if k == 'gemm_f32_t128_kernel' else k
    return 'gemm_h2_kernel' if k == 'gemm_h2m16_kernel' else k          # the two MFMA shapes of the h2 product: one class in bench.py


def load(sub):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (d, sub), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = name(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r.get('Dispatch_Id'), r.get('Agent_Id'))
        if key not in disp[k]:
            disp[k].add(key)
            if r.get('End_Timestamp') and r.get('Start_Timestamp'):
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return agg, {k: len(v) for k, v in disp.items()}, dur


busy, nb, durb = load('busy')
mops, nm, _ = load('mops')
out = {}
for k in sorted(busy, key=lambda k: -busy[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
    b = busy[k]
    if b.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
        continue
    n = nb[k]
    gui = b.get('GRBM_GUI_ACTIVE', 0) / 8.0
    e = dict(launches=n, avg_us=round(durb[k] / n, 2) if durb.get(k) else None,
             mfma_busy_cycles_per_launch=round(b['SQ_VALU_MFMA_BUSY_CYCLES'] / n),
             gpu_cycles_per_launch=round(gui / n),
             mfma_busy_frac=round(b['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024), 4) if gui else None)
    m = mops.get(k, {})
    if m:
        e.update(mfma_mops_f32_per_launch=round(m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) / max(nm[k], 1)),
                 mfma_mops_f16_per_launch=round(m.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0) / max(nm[k], 1)),
                 mfma_insts_per_launch=round(m.get('SQ_INSTS_MFMA', 0) / max(nm[k], 1)),
                 valu_insts_per_launch=round(m.get('SQ_INSTS_VALU', 0) / max(nm[k], 1)))
    out[k] = e
out['_commit'] = sys.argv[2] if len(sys.argv) > 2 else None
print(json.dumps(out, indent=1))
