import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench, echr_amd
from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
from echr_amd.optim import ClampAdam
dev = torch.device('cuda', 0)
opt, params, vid = bench.make_workload(0, False)
model = echr_amd.CaptionGenerator(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).train()
crit = LanguageModelCriterion()
arena = model.build_arena()
optim = ClampAdam(model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=arena)
tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
labels = torch.from_numpy(vid['labels'])
tgt = labels[:, 1:].to(dev); msk = torch.from_numpy(vid['masks'])[:, 1:].to(dev)
fused = None
if 'autograd' not in sys.argv:          # default: the one-call iteration (echr_train_step); `autograd` measures the autograd path
    from echr_amd.fused import FusedTrainStep
    fused = FusedTrainStep(model, optim, grad_clip=opt.grad_clip)
def iteration():
    if fused is not None:
        return fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt, msk)
    optim.zero_grad()
    pred = model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
    loss = crit(pred, tgt, msk)
    loss.backward()
    clip_gradient(optim, opt.grad_clip)
    optim.step()
for _ in range(5): iteration()
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K): iteration()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.3f ms/iter, total %.3f ms/iter, GPU drain after last enqueue %.3f ms' % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
# host-only cost: same loop while the GPU is idle at start of each iteration (sync each iteration)
ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); iteration(); ts.append(time.perf_counter() - a)
print('host time of one iteration issued into an empty queue: min %.3f ms' % (min(ts) * 1e3))
if 'profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20):
        iteration()
        torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
    st.sort_stats('tottime').print_stats(30)
if 'calls' in sys.argv:
    # time spent inside each library entry point (host side of the ctypes call: argument marshalling + the HIP launches it issues)
    from echr_amd import _lib
    lib = _lib.load()
    acc = {}
    import threading
    lock = threading.Lock()
    def wrap(name, fn):
        def f(*a):
            t = time.perf_counter()
            r = fn(*a)
            dt = time.perf_counter() - t
            with lock:
                e = acc.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += dt
            return r
        return f
    for name, _, _ in _lib.SYMBOLS:
        if name not in ('echr_last_error', 'echr_version'):
            setattr(lib, name, wrap(name, getattr(lib, name)))
    R = 20
    for _ in range(R):
        iteration(); torch.cuda.synchronize()
    tot = 0.0
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print('%-32s %5.1f calls/iter %8.1f us/iter' % (k, n / R, t / R * 1e6)); tot += t / R * 1e6
    print('library calls total %.1f us/iter' % tot)
