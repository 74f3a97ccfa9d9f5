#!/bin/bash
# HBM traffic per kernel from the L2 memory-side counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (they do not fit one pass); units KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B,
# so reads are doubled by tools/pmc_traffic_summary.py.
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_traffic}; shift          # further arguments go to bench.py (e.g. --c5)
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --regions 1 --no-others --no-cpu --no-roofline --no-native "$@" > $out/fetch.log 2>&1; echo fetch_exit=$?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --regions 1 --no-others --no-cpu --no-roofline --no-native "$@" > $out/write.log 2>&1; echo write_exit=$?
