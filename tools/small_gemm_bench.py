import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as G
from echr_amd import _lib
lib = _lib.load()
for name, lay, M, N, K in [('q3', 'NT', 64, 512, 512), ('gate', 'NT', 4096, 16, 512), ('ev0', 'NT', 64, 2048, 512), ('wg_ev', 'TN', 2048, 512, 64),
                           ('dx_ev', 'NN', 64, 512, 2048), ('emb', 'NT', 64, 512, 1012), ('g_fc2', 'TN', 16, 512, 4096), ('dP1', 'NN', 4096, 512, 16),
                           ('wg_emb', 'TN', 512, 1012, 64)]:
    for sp in (0, 1, 2, 4, 8):
        lib.echr_config_set(b'gemm_split', sp)
        us, tf = G.run(name, lay, M, N, K, reps=50)
        print('%-8s %s %5d %5d %5d split=%d  %6.1f us' % (name, lay, M, N, K, sp, us), flush=True)
