python -m pytest tests -x -q -m gpu -k "h2 or gemm or pack" 2>&1 | tail -3
H2_ONLY=g_w_hh,g_w_c2a,dOUT_c python tools/h2_bench.py 2>/dev/null
(cd .ab_base; H2_ONLY=g_w_hh,g_w_c2a,dOUT_c python tools/h2_bench.py 2>/dev/null)
bash tools/ab_rounds.sh
