import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, echr_amd
from echr_amd import synth, _lib
lib = _lib.load()
dev = torch.device('cuda')
opt = synth.default_opt(vocab_size=5000, seq_length=19)
params = synth.make_params(opt, 0)
m = echr_amd.CaptionGenerator(opt)
m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
m = m.to(dev).eval()
for N, T_v in ((64, 8192), (1000, 256)):
    vid = synth.make_video(N, 128, 21, 5001, seed=7, T_v=T_v if N > 64 else None, full_len=(N == 64), disjoint=(N == 64))
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    for k in (0, 11, 6):
        lib.echr_config_set(b'persist_sample_force_eos', k)
        with torch.no_grad():
            for _ in range(2):
                seq, lp = m(tap, c3d, lda, [], vid['ind'], vid['soi'], mode='eval')
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                seq, lp = m(tap, c3d, lda, [], vid['ind'], vid['soi'], mode='eval')
            torch.cuda.synchronize()
        print('N=%d force_eos=%d: %d columns, %.2f ms per decode' % (N, k, seq.shape[1], (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    lib.echr_config_set(b'persist_sample_force_eos', 0)
