#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB), with the gfx950 correction
(FETCH_SIZE reports exactly half of the bytes of wide coalesced reads -> doubled).  Writes JSON: kernel -> bytes/launch."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]


def load(sub, counter):
    f = glob.glob('%s/%s/**/*counter_collection.csv' % (d, sub), recursive=True)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('echr::', '')
        k = k.split('<')[0] if k.startswith(('gemm_f32_kernel', 'gemm_h2_kernel', 'gemm_h2m16_kernel', 'dec_persist', 'att_post_kernel')) else k
        k = 'gemm_h2_kernel' if k == 'gemm_h2m16_kernel' else k          # the two MFMA shapes of the h2 product: one class in bench.py
        k = 'gemm_f32_kernel' if k.startswith('gemm_f32_t128_kernel') else k          # the two tile sizes of the exact-fp32 product: one class too
        agg[k] += float(r['Counter_Value'])
        cnt[k] += 1
    return agg, cnt


fe, nf = load('fetch', 'FETCH_SIZE')
wr, nw = load('write', 'WRITE_SIZE')
out = {}
for k in sorted(fe, key=lambda k: -fe[k]):
    if not (k.startswith(('gemm', 'rec_gemm', 'att_', 'lstm', 'clamp_adam', 'h2_pack', 'dec_persist', 'sst_persist'))):
        continue
    out[k] = dict(launches=nf[k], fetch_bytes_per_launch=round(2 * 1024 * fe[k] / nf[k]), write_bytes_per_launch=round(1024 * wr.get(k, 0) / max(nw.get(k, 1), 1)))
    out[k]['hbm_bytes_per_launch'] = out[k]['fetch_bytes_per_launch'] + out[k]['write_bytes_per_launch']
# the four persistent recurrence kernels run as two concurrent pairs (forward, reverse); bench.py times each pair as one launch
pk = [k for k in out if k.startswith('dec_persist')]
if pk:
    pairs = max(1, max(out[k]['launches'] for k in pk))
    out['dec_persist_kernels'] = dict(launches=2 * pairs, members=pk,
                                      fetch_bytes_per_launch=round(sum(out[k]['fetch_bytes_per_launch'] * out[k]['launches'] for k in pk) / (2 * pairs)),
                                      write_bytes_per_launch=round(sum(out[k]['write_bytes_per_launch'] * out[k]['launches'] for k in pk) / (2 * pairs)))
    out['dec_persist_kernels']['hbm_bytes_per_launch'] = out['dec_persist_kernels']['fetch_bytes_per_launch'] + out['dec_persist_kernels']['write_bytes_per_launch']
# the proposal encoder's persistent pair (forward, reverse): one class in bench.py, two launches per iteration
sk = [k for k in out if k.startswith('sst_persist')]
if sk:
    n = sum(out[k]['launches'] for k in sk)
    out['sst_persist_kernels'] = dict(launches=n, members=sk,
                                      fetch_bytes_per_launch=round(sum(out[k]['fetch_bytes_per_launch'] * out[k]['launches'] for k in sk) / n),
                                      write_bytes_per_launch=round(sum(out[k]['write_bytes_per_launch'] * out[k]['launches'] for k in sk) / n))
    out['sst_persist_kernels']['hbm_bytes_per_launch'] = out['sst_persist_kernels']['fetch_bytes_per_launch'] + out['sst_persist_kernels']['write_bytes_per_launch']
out['_commit'] = sys.argv[2] if len(sys.argv) > 2 else None          # the tree these counters were collected from
print(json.dumps(out, indent=1))
