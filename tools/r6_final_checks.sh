#!/bin/bash
# smoke(), the tests added last, and the fabric traffic of the recurrence pairs with XCD-aware roles
cd $GRAFT_REPO_ROOT; out=gpurun_out/r6o; mkdir -p $out
timeout -k 10 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo smoke_exit=$?; tail -2 $out/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -q -k "t128 or initial_state_map or rehearsal" > $out/tests.log 2>&1; echo test_exit=$?; tail -3 $out/tests.log
export ECHR_PERSIST_XCD=1
bash tools/pmc_traffic.sh r6o_xcd_raw > $out/pmc_xcd.log 2>&1
python3 tools/pmc_traffic_summary.py gpurun_out/r6o_xcd_raw xcd1 > $out/pmc_traffic_xcd1.json; rm -rf gpurun_out/r6o_xcd_raw
python3 -c "
import json
a=json.load(open('$out/pmc_traffic_xcd1.json')); b=json.load(open('profiles/r06_pmc_traffic.json'))
for k in ('dec_persist_kernels','dec_persist_fwd_kernel','dec_persist_bwd_kernel'):
    if k in a and k in b: print(k, 'xcd-aware', a[k]['hbm_bytes_per_launch'], 'plain', b[k]['hbm_bytes_per_launch'])
"
