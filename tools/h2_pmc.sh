#!/bin/bash
# L2 hit/miss + HBM fetch of the h2 GEMM on the logits shape (one rocprofv3 --pmc pass per counter group)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-h2pmc}
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tools')
import h2_bench as HB, torch, ctypes as C
from echr_amd import _lib as L
for name, M, N, K in (('logits', 1280, 5001, 1536), ('gin_x3', 3840, 2048, 512), ('g_w_logit', 5001, 1536, 1280)):
    A, B = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
    Ax, Bx = HB.pack(A), HB.pack(B)
    Cc = torch.zeros(M, N, device='cuda')
    d = HB.desc(Ax, Bx, Cc, M, N, K, split=1)
    for _ in range(5): HB.lib.echr_gemm_f32(C.byref(d), L.stream_ptr())
    torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/p1 -- python3 /tmp/one.py > $out/run1.log 2>&1
echo exit=$?
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/p2 -- python3 /tmp/one.py > $out/run2.log 2>&1
echo exit=$?
python3 - <<PY
import csv, glob, collections
for p in ('p1', 'p2'):
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % p, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'gemm_h2' in r['Kernel_Name']]
        byd = collections.defaultdict(dict)
        for r in rows: byd[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        for i, (k, v) in enumerate(sorted(byd.items(), key=lambda kv: int(kv[0]))):
            if i % 5 == 4: print(p, k, {a: int(b) for a, b in v.items()})
PY
