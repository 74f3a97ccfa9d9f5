#!/usr/bin/env python3
"""bench.py -- caption-decoder timesteps/sec (fwd+bwd+clamp+Adam) on N MI355X, with roofline and CPU baseline.

One "step" = one training iteration of the caption hot path over one video batch:
  CaptionGenerator.forward(mode='train') -> LanguageModelCriterion -> backward -> [all-reduce SUM] -> clip_gradient + Adam.
Workload (BASELINE.json configs[2], "c3"): N=64 events x A=128 segments x 500-d C3D features, S=20 decoder timesteps,
V1=5001 (the reference does not fix the vocabulary; stated here), fp32, synthetic N(0,1) features, random-init weights.
The 64 events are laid out disjointly on a T_v = 8192 video so that all 64x128 feature rows are distinct
(--overlap uses SURVEY 8-d's T_v = 160 one-video layout instead, where events share rows).
metric = decoder timesteps/s = (iterations x S x n_gpus) / wall time; every rank works on its own video (weak scaling).

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--overlap] [--no-cpu] [--no-roofline]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

S_STEPS, N_EV, A_SEG, V1 = 20, 64, 128, 5001
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (same guide; the 2:1-sparsity figure is not used)


def make_workload(rank, overlap, c5=False):
    from echr_amd import synth
    opt = synth.default_opt(vocab_size=V1 - 1, seq_length=S_STEPS - 1)
    params = synth.make_params(opt, 0)
    if c5:      # BASELINE config 5: one 256-segment video, proposals of 4..256 segments, SST in the loop
        vid = synth.make_video(N_EV, 256, S_STEPS + 1, V1, seed=1234 + rank, T_v=256)
        rs = np.random.RandomState(99 + rank)
        vid['tap_labels'] = (rs.uniform(size=(256, opt.K)) > 0.9).astype(np.float32)
        vid['tap_masks'] = (np.arange(256)[:, None] >= np.arange(opt.K)[None, :]).astype(np.float32)
        vid['w1'] = rs.uniform(0.05, 0.3, size=(opt.K,)).astype(np.float32)
        return opt, params, vid
    vid = synth.make_video(N_EV, A_SEG, S_STEPS + 1, V1, seed=1234 + rank, full_len=True, disjoint=not overlap,
                           T_v=None if not overlap else 160)
    return opt, params, vid


def gpu_leg(args, rank, world, local_rank):
    import echr_amd
    from echr_amd import _lib, parallel
    from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
    from echr_amd.optim import ClampAdam
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    use_dist = dist.is_available() and dist.is_initialized()
    opt, params, vid = make_workload(rank, args.overlap, args.c5)
    model = echr_amd.CaptionGenerator(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev).train()
    crit = LanguageModelCriterion()
    arena = None if args.no_arena else model.build_arena()      # flat parameter/gradient buffers: 1-launch Adam, 1-bucket all-reduce
    if use_dist and arena is not None and os.environ.get('ECHR_DP_OVERLAP', '1') != '0':
        parallel.enable_overlap(model)       # logit-layer gradients (35 % of the bytes) are reduced while the reverse recurrence runs
    optim = ClampAdam(model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=arena)
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])                       # host copy: step count needs no device sync
    tgt = labels[:, 1:].to(dev)
    msk = torch.from_numpy(vid['masks'])[:, 1:].to(dev)
    tgt_h, msk_h = labels[:, 1:].numpy(), vid['masks'][:, 1:]

    if args.c5:
        from echr_amd import models as EM
        from echr_amd.misc.utils import TAPModelCriterion
        torch.manual_seed(0)
        tap_model = EM.setup_tap(opt).to(dev)
        tap_model.train()
        tap_arena = None if args.no_arena else tap_model.build_arena()
        tap_optim = ClampAdam(tap_model.parameters(), lr=opt.lr, arena=tap_arena)
        tap_crit = TAPModelCriterion()
        tl, tm, tw = (torch.from_numpy(vid[k]).to(dev) for k in ('tap_labels', 'tap_masks', 'w1'))

    def c5_iteration():      # train.py's joint 'tap_cg' iteration: SST -> caption path -> lambda1*tap_loss + lambda2*cg_loss
        if fused is not None:
            # the caption side as ONE library call (forward, criterion, backward, clip, Adam); d loss / d tap_feats comes back in g_tap and is
            # backpropagated into the proposal encoder together with its own loss (same sums as loss.backward() of the joint loss)
            tap_optim.zero_grad()
            early = os.environ.get('ECHR_EARLY_PREPARE', '1') != '0'
            if early:          # the caption side's tap-independent part starts now and runs beside the proposal encoder's forward
                fused.prepare(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
            tap_feats, props = tap_model(c3d)
            g_tap = torch.zeros_like(tap_feats)
            tap_loss = 0.01 * tap_crit(props, tm, tl, tw)          # (queued before the caption side: nothing of it waits for g_tap)
            defer = os.environ.get('ECHR_DEFER_UPDATE', '1') != '0'
            cg_loss = fused(tap_feats.detach(), c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tap_grad=g_tap,
                            defer_update=defer, prepared=early)
            torch.autograd.backward([tap_loss, tap_feats], [None, g_tap])
            clip_gradient(tap_optim, opt.grad_clip)
            tap_optim.step()
            return tap_loss.detach() + cg_loss
        optim.zero_grad()
        tap_optim.zero_grad()
        tap_feats, props = tap_model(c3d)
        pred = model(tap_feats, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
        loss = 0.01 * tap_crit(props, tm, tl, tw) + crit(pred, tgt, msk)
        loss.backward()
        if use_dist:
            parallel.allreduce_gradients(model, force=True)
            parallel.allreduce_gradients(tap_model, force=True)
        clip_gradient(optim, opt.grad_clip)
        clip_gradient(tap_optim, opt.grad_clip)
        optim.step()
        tap_optim.step()
        return loss

    def fwd_only():
        with torch.no_grad():
            pred = model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
            return crit(pred, tgt, msk)

    # the whole iteration as ONE library call (echr_train_step: include/echr_hip.h, echr_amd/fused.py) instead of ~75 ctypes calls and four
    # autograd nodes; single-process default.  With several ranks the autograd path keeps the staged early all-reduce, unless --fused asks
    # for the one-call form there too (backward inside the call, then ONE collective, then clip + step)
    fused = None
    want_fused = args.fused == 'on' or (args.fused == 'auto' and not use_dist)
    if want_fused and arena is not None and not (args.c5 and use_dist):
        from echr_amd.fused import FusedTrainStep
        fused = FusedTrainStep(model, optim, grad_clip=opt.grad_clip)

    def iteration():
        if args.c5:
            return c5_iteration()
        if args.mode == 'fwd':
            if fused is not None:          # forward + criterion only (BASELINE config 2), same one-call entry
                return fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, forward_only=True)
            return fwd_only()
        if fused is not None:
            # (criterion inputs handed over on the host, as the reference's loader produces them: they travel with the index vectors)
            if not use_dist:
                return fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
            loss = fused(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False)
            parallel.allreduce_gradients(model, force=True)
            clip_gradient(optim, opt.grad_clip)
            optim.step()
            return loss
        optim.zero_grad()
        pred = model(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
        loss = crit(pred, tgt, msk)
        loss.backward()
        if use_dist:
            parallel.allreduce_gradients(model, force=True)        # SUM over ranks == reference m_batch accumulation
        clip_gradient(optim, opt.grad_clip)
        optim.step()
        return loss

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def say(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    say('model ready; warm-up')
    for i in range(args.warmup):
        loss = iteration()
        torch.cuda.synchronize()
        say('warm-up %d done' % i)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = iteration()
    t_issue = time.perf_counter() - t0          # host time to issue the timed steps (the GPU may still be working)
    fence()
    dt = time.perf_counter() - t0
    say('host issue time %.3f ms per step (GPU-inclusive %.3f)' % (1e3 * t_issue / args.steps, 1e3 * dt / args.steps))
    if os.environ.get('ECHR_HOST_PROBE') == '1':          # diagnostic: host time of one iteration issued into an empty queue
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t1 = time.perf_counter(); iteration(); ts.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        say('host time of one iteration issued into an empty queue: min %.3f ms, median %.3f ms' % (1e3 * min(ts), 1e3 * sorted(ts)[5]))
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    say('timed region done: %.3f s for %d steps' % (dt, args.steps))

    # second timed figure with every large projection on the NATIVE fp32 MFMA path (no fp16-pair / bf16-plane emulation)
    native = None
    if not args.no_native and args.mode == 'train' and not args.c5:
        lib = _lib.load()
        lib.echr_config_set(b'gemm_h2', 0)
        lib.echr_config_set(b'persist_h2', 0)
        lib.echr_config_set(b'gemm_bf16x3', 0)
        for _ in range(2):
            iteration()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            iteration()
        fence()
        dtn = time.perf_counter() - t0
        lib.echr_config_set(b'gemm_h2', 1)
        lib.echr_config_set(b'persist_h2', 1)
        lib.echr_config_set(b'gemm_bf16x3', 1)
        if use_dist:
            t = torch.tensor([dtn], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtn = float(t.item())
        native = dict(value=round(args.steps * S_STEPS * world / dtn, 1), unit='timesteps/s', ms_per_step=round(1e3 * dtn / args.steps, 3),
                      note='same workload with gemm_h2=0, gemm_bf16x3=0, persist_h2=0: every product on v_mfma_f32_* (exact fp32 MFMA)')
        for _ in range(2):
            iteration()
        fence()

    roof = None
    if not args.no_roofline:
        # second, instrumented pass over the same iterations: HIP events around every launch of each kernel class.  EVERY rank runs
        # the iterations (they contain the gradient collectives); only rank 0 instruments and reports.
        lib = _lib.load()
        if rank == 0:
            lib.echr_prof_enable(1)
        for _ in range(max(2, min(args.steps, 5))):
            iteration()
        fence()
    if rank == 0 and not args.no_roofline:
        # kernel classes of libechr_hip.so (echr_prof_read kinds) -> (name, rocprof kernel symbol, bound)
        kinds = {0: ('gemm_f32_kernel', 'gemm_f32_kernel<*> (all tile/layout instantiations)', 'mfma'),
                 7: ('gemm_h2_kernel', 'gemm_h2m16_kernel<2> (v_mfma_f32_16x16x32_f16) + gemm_h2_kernel<128, 32, 2> (many-way split-K products)', 'mfma'),
                 6: ('gemm_split_kernel', 'gemm_split_kernel', 'mfma'),
                 4: ('rec_gemm_kernel', 'rec_gemm_kernel', 'mfma'),
                 8: ('h2_pack_kernel', 'h2_pack_kernel', 'hbm'),
                 1: ('att_fwd', 'att_score_kernel + att_context_kernel', 'hbm'),
                 2: ('att_bwd_kernel', 'att_bwd_kernel', 'hbm'),
                 3: ('att_post_kernel', 'att_post_kernel', 'valu'),
                 9: ('dec_persist_kernels', 'dec_persist_fwd_kernel<true, BIG> (forward) and dec_persist_bwd_kernel<BIG> (reverse; BIG = events longer than 129 segments): each ONE launch of 256 workgroups '
                                            '(attention chain on 192, the two plain LSTM streams on 64) covering all S steps', 'mfma'),
                 10: ('sst_persist_kernels', 'sst_persist_fwd_kernel / sst_persist_bwd_kernel: the proposal encoder\'s two-layer LSTM over the video, ONE '
                                             'launch of 64 workgroups per direction (recurrent matrices in registers, batch-1 GEMV chain: latency-bound)', 'hbm')}
        n_it = max(2, min(args.steps, 5))
        stats = {}
        for k, (name, sym, bound) in kinds.items():
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            lib.echr_prof_read(k, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n))
            stats[name] = dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value, sym=sym, bound=bound)
        lib.echr_prof_enable(0)
        # HBM traffic per launch cannot be collected inside a timed run (PMC needs rocprofv3 --pmc in separate passes): it is read from
        # the committed summary of the SAME command (tools/pmc_traffic.sh -> profiles/r02_pmc_traffic.json); null when absent
        # (one pair of files per workload: `_c5` for --c5; other variants -- overlapping rows, forward only -- carry no counters)
        sfx = '_c5' if args.c5 else ''
        traffic, traffic_src, mfma_pmc = {}, None, {}
        if not (args.overlap or args.mode != 'train'):
            for name in ['r04_pmc_traffic%s.json' % sfx] + ([] if args.c5 else ['r03_pmc_traffic.json', 'r02_pmc_traffic.json']):
                try:
                    traffic = json.load(open(os.path.join(ROOT, 'profiles', name)))
                    traffic_src = 'profiles/' + name
                    break
                except Exception:
                    pass
            # MFMA activity (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD-cycles of the dispatch) likewise comes from the committed --pmc pass of the
            # same command (tools/pmc_mfma.sh -> profiles/r04_pmc_mfma.json)
            for name in ['r04_pmc_mfma%s.json' % sfx] + ([] if args.c5 else ['r03_pmc_mfma.json', 'r02_pmc_mfma.json']):
                try:
                    mfma_pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
                    break
                except Exception:
                    pass
        traffic_commit = traffic.get('_commit') if isinstance(traffic, dict) else None

        def mfma_busy(name):
            if name == 'dec_persist_kernels':           # time-weighted over the four kernels of the two pairs
                ks = [v for k, v in mfma_pmc.items() if k.startswith('dec_persist') and isinstance(v, dict) and v.get('mfma_busy_frac') is not None]
                tot = sum(v['avg_us'] * v['launches'] for v in ks)
                return round(sum(v['mfma_busy_frac'] * v['avg_us'] * v['launches'] for v in ks) / tot, 4) if tot else None
            v = mfma_pmc.get(name)
            return v.get('mfma_busy_frac') if isinstance(v, dict) else None

        # HIP-event pairs around a short launch add a fixed cost per launch (the second record waits for the first to retire):
        # measured here on an idle stream and subtracted, so that avg_launch_us agrees with rocprofv3's kernel durations
        ev_ms, ev_n = C.c_double(), C.c_int64()
        lib.echr_prof_event_overhead(C.byref(ev_ms), C.byref(ev_n))
        ev_us = 1e3 * ev_ms.value / max(ev_n.value, 1)

        def line(name):
            st = stats[name]
            if st['launches'] == 0 or st['ms'] <= 0:
                return None
            ms = max(st['ms'] - 1e-3 * ev_us * st['launches'], 1e-6)
            if st['bound'] == 'mfma':
                # fp32-grade products: native fp32 MFMA, or 3 fp16 (h2) / 6 bf16 (split) MFMA products per product -> the ceiling of
                # the ALGORITHMIC rate is the 16-bit dense MFMA peak divided by the products spent per fp32-grade product
                peak = {'gemm_h2_kernel': MFMA_16BIT_PEAK_TFLOPS / 3, 'gemm_split_kernel': MFMA_16BIT_PEAK_TFLOPS / 6}.get(name, MFMA_F32_PEAK_TFLOPS)
                ach, peak, unit = st['flops'] / (ms * 1e-3) / 1e12, round(peak, 1), 'TFLOP/s'
            elif st['bound'] == 'valu':
                # transcendental-bound pass: 6 flops per (slot, feature, timestep) counted against the fp32 vector peak (= the fp32 MFMA peak)
                ach, peak, unit = st['flops'] / (ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, 'TFLOP/s'
            else:
                ach, peak, unit = st['bytes'] / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, 'GB/s'
            tr = traffic.get(name) or next((v for k, v in traffic.items() if k.startswith(name + '<')), None)
            tr = tr if isinstance(tr, dict) else None
            tb = (tr or {}).get('hbm_bytes_per_launch')
            alg = st['bytes'] / st['launches'] if st['bytes'] > 0 else None
            return dict(bound='mfma' if st['bound'] == 'valu' else st['bound'], kernel=st['sym'], achieved=round(ach, 2), peak=peak, unit=unit,
                        frac=round(ach / peak, 4), traffic=tb, traffic_source=traffic_src if tb else None, traffic_commit=traffic_commit if tb else None,
                        algorithmic_bytes_per_launch=round(alg) if alg else None,
                        traffic_over_algorithmic=round(tb / alg, 2) if (tb and alg) else None,
                        mfma_busy_frac_pmc=mfma_busy(name),
                        avg_launch_us=round(1e3 * ms / st['launches'], 2), event_overhead_us_subtracted=round(ev_us, 2),
                        launches_per_step=st['launches'] / n_it, ms_per_step=round(ms / n_it, 3))

        dom = max(stats, key=lambda k: stats[k]['ms'])
        roof = line(dom)
        # `achieved` = algorithmic flops (2MNK) or bytes per launch / HIP-event duration on the launch stream; `peak` for the
        # MFMA-bound kernels: 157.3 TF (native fp32 MFMA: gemm_f32, rec_gemm), 2500/3 TF (gemm_h2: three fp16 MFMA products per
        # fp32-grade product), 2500/6 TF (gemm_split: six bf16 products).
        roof['other_kernels'] = {k: line(k) for k in stats if k != dom and line(k) is not None}
    host_path = 'echr_train_step (one library call per iteration)' if fused is not None else 'autograd Functions (one ctypes call per fused region)'
    return dt, final_loss, roof, native, host_path


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_leg(args):
    """The oracle (a CPU port of the reference algorithm, kind='port') timed on this box's host cores on the SAME workload
    (fwd + criterion + backward + clamp + Adam), per BASELINE.md section 3: all cores available to this process AND one thread,
    2 warm-ups, min of up to 5 repeats, bounded to ~25 s per setting; the forward-only figure (BASELINE config 2) beside it.
    --c5: the joint SST + caption iteration (oracle sst_forward + tap_criterion + caption path)."""
    from oracle import echr_ref_cpu as O
    opt, params, vid = make_workload(0, args.overlap, args.c5)
    # host cores actually available to this process: the scheduler affinity, capped by the cgroup CPU quota when one is set
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except Exception:
        pass
    if os.environ.get('ECHR_CPU_THREADS'):
        avail = max(1, min(avail, int(os.environ['ECHR_CPU_THREADS'])))
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
    if args.c5:      # the proposal encoder's parameters (nn.LSTM layout), same init ranges as the GPU leg's module
        H, D, K = opt.hidden_dim, opt.video_dim, opt.K
        rs0 = np.random.RandomState(7)
        shapes = {'rnn.weight_ih_l0': (4 * H, D), 'rnn.weight_hh_l0': (4 * H, H), 'rnn.bias_ih_l0': (4 * H,), 'rnn.bias_hh_l0': (4 * H,),
                  'rnn.weight_ih_l1': (4 * H, H), 'rnn.weight_hh_l1': (4 * H, H), 'rnn.bias_ih_l1': (4 * H,), 'rnn.bias_hh_l1': (4 * H,),
                  'scores.weight': (K, H), 'scores.bias': (K,)}
        for k, shp in shapes.items():
            P['tap.' + k] = torch.from_numpy((rs0.uniform(-1, 1, size=shp) / np.sqrt(H)).astype(np.float32)).requires_grad_(True)
    ms = {k: torch.zeros_like(v) for k, v in P.items()}
    vs = {k: torch.zeros_like(v) for k, v in P.items()}
    tap, c3d, lda = (torch.from_numpy(vid[k]) for k in ('tap', 'c3d', 'lda'))
    labels, masks = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    if args.c5:
        tl, tm, tw = (torch.from_numpy(vid[k]) for k in ('tap_labels', 'tap_masks', 'w1'))
    rs = np.random.RandomState(0)

    def drop(site, step, shape):
        p = 0.3 if site == 'tsrm' else 0.5
        return torch.from_numpy(((rs.random_sample(shape) >= p) / (1 - p)).astype(np.float32))

    def forward():
        tap_in, loss = tap, 0.0
        if args.c5:
            Pt = {k[4:]: v for k, v in P.items() if k.startswith('tap.')}
            tap_in, props = O.sst_forward(Pt, c3d, drop('sst', 0, (c3d.shape[0], opt.hidden_dim)))
            loss = 0.01 * O.tap_criterion(props, tm, tl, tw)
        pred = O.caption_forward(P, tap_in, c3d, lda, labels, vid['ind'], vid['soi'], 'train', drop, opt.n_head)
        return loss + O.lm_criterion(pred, labels[:, 1:], masks[:, 1:])

    def iteration(step, fwd_only=False):
        if fwd_only:
            with torch.no_grad():
                forward()
            return
        for v in P.values():
            v.grad = None
        forward().backward()
        with torch.no_grad():
            for k, v in P.items():
                if v.grad is not None:
                    O.clamp_adam_step(v, v.grad, ms[k], vs[k], step, opt.lr, clip=opt.grad_clip)

    def timed(threads, budget_s, warm, fwd_only=False):
        torch.set_num_threads(threads)
        print('[bench] cpu baseline: %d thread(s)%s: %d warm-up(s)' % (threads, ' forward only' if fwd_only else '', warm), file=sys.stderr, flush=True)
        step = 1
        t_all = time.perf_counter()
        for _ in range(warm):
            iteration(step, fwd_only)
            step += 1
            if time.perf_counter() - t_all > budget_s:
                break
        times = []
        while len(times) < 5 and (not times or time.perf_counter() - t_all < budget_s):
            t0 = time.perf_counter()
            iteration(step, fwd_only)
            times.append(time.perf_counter() - t0)
            step += 1
        return times

    t_all = timed(avail, 25.0, 2)
    t_fwd = timed(avail, 10.0, 1, fwd_only=True)
    t_one = timed(1, 25.0, 1 if avail > 1 else 0) if avail > 1 else t_all
    best = min(t_all)
    what = ('joint SST + caption iterations of the same c5 workload (T_v=256, 64 proposals of 4..256 segments, S=20)' if args.c5 else
            'fwd+bwd+clamp+Adam iterations of the same N=64 x A=128 x S=20 workload')
    return dict(value=round(S_STEPS / best, 2), unit='timesteps/s', cores=avail, kind='port',
                median=round(S_STEPS / float(np.median(t_all)), 2),
                fwd_only=dict(value=round(S_STEPS / min(t_fwd), 2), unit='timesteps/s', repeats=len(t_fwd)),
                one_thread=dict(value=round(S_STEPS / min(t_one), 2), repeats=len(t_one)),
                cpu_model=cpu_model_name(), torch=torch.__version__, os_cpu_count=os.cpu_count(),
                sample='%s: min of %d after 2 warm-ups on all %d available cores (os.cpu_count()=%s), forward-only min of %d, and min of %d on '
                       '1 thread; torch %s CPU fp32' % (what, len(t_all), avail, os.cpu_count(), len(t_fwd), len(t_one), torch.__version__))


def self_launch(args):
    """Start `--gpus N` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and relay the
    child's stdout (rank 0 prints the one JSON line) and exit code.  Runs before this process has made any GPU call."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    print('[bench] launching %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)       # 0.2 s of timed work: one host hiccup no longer moves the average
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--overlap', action='store_true', help='SURVEY 8-d one-video layout (T_v=160, events share rows)')
    ap.add_argument('--mode', choices=['train', 'fwd'], default='train',
                    help="'train' = fwd+bwd+clamp+Adam (BASELINE config 3, the headline metric); 'fwd' = forward + loss only (config 2)")
    ap.add_argument('--c5', action='store_true', help='BASELINE config 5: SST proposal encoder over a 256-segment video + caption '
                    'path with proposals of up to 256 segments, joint fwd+bwd+Adam (extra line; the headline metric is the default)')
    ap.add_argument('--no-arena', action='store_true', help='per-tensor gradients/optimiser instead of the flat arena')
    ap.add_argument('--fused', choices=['auto', 'on', 'off'], default=os.environ.get('ECHR_BENCH_FUSED', 'auto'),
                    help='one echr_train_step call per iteration instead of the autograd path (auto: on for one process, off with several ranks)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-native', action='store_true', help='skip the second timed figure on the native fp32 MFMA path')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: start the N ranks ourselves (one process per GPU) BEFORE anything in this
        # process touches the GPU -- the child is torch.distributed.run, this process only relays its output and exit code
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('ECHR_BENCH_BACKEND', 'nccl')            # 'gloo': rehearsal of the N-rank path on a 1-GPU box
    if os.environ.get('ECHR_BENCH_ONE_GPU') == '1':                   # rehearsal: every rank on cuda:0 (needs the gloo backend)
        assert backend == 'gloo', 'ECHR_BENCH_ONE_GPU=1 needs ECHR_BENCH_BACKEND=gloo (RCCL wants one device per rank)'
        local_rank = 0
    if world > 1 or os.environ.get('ECHR_FORCE_DIST') == '1':       # ECHR_FORCE_DIST: exercise the RCCL path with one rank
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get('ECHR_BENCH_DRYRUN') == '1':
        # launcher check without a GPU (tests/test_parallel_gloo.py): rendezvous, one MAX all-reduce like the timing reduction, one JSON line
        assert backend == 'gloo' and world == args.gpus
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        if rank == 0:
            print(json.dumps({'dryrun': True, 'n_gpus': world, 'max_rank_plus_1': float(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus %d`, which launches the ranks itself, or under '
                 'torch.distributed.run --nproc-per-node %d)' % (args.gpus, world, args.gpus, args.gpus))
    dp_info = {}
    if world > 1:
        # several ranks may share CUs with collective kernels (and, in the one-GPU rehearsal, with each other): the persistent recurrences
        # are launched cooperatively, so a grid only starts once all of its workgroups can be resident
        # -- but only where something else CAN hold CUs while a persistent pair runs: ranks sharing one device (the rehearsal).  With one
        # device per rank every collective of an iteration is behind the reverse recurrence (EarlyReducer: `defer_first`, hand-over after
        # the recurrence) and is waited for before clamp + Adam, i.e. before the next forward pair in stream order, so no collective
        # kernel is ever resident beside a pair and the plain launch (25 us per pair cheaper) is safe.
        from echr_amd import _lib
        shared_device = os.environ.get('ECHR_BENCH_ONE_GPU') == '1'
        coop = int(os.environ.get('ECHR_PERSIST_COOP', '1' if shared_device or os.environ.get('ECHR_DP_DEFER_FIRST', '1') == '0' else '0'))
        _lib.load().echr_config_set(b'persist_coop', coop)
        dp_info.update(persist_coop=coop, dp_algo=__import__('echr_amd.parallel', fromlist=['choose_algo']).choose_algo(world),
                       dp_overlap=os.environ.get('ECHR_DP_OVERLAP', '1') != '0' and args.fused != 'on')
        if os.environ.get('ECHR_BENCH_ONE_GPU') == '1' and os.environ.get('ECHR_BENCH_PERSIST', '0') == '0':
            # rehearsal with every rank on ONE device: two 256-workgroup persistent grids must not share it -> launch-per-phase recurrences
            _lib.load().echr_config_set(b'persist', 0)
            _lib.load().echr_config_set(b'persist_bwd', 0)
    dt, loss, roof, native, host_path = gpu_leg(args, rank, world, local_rank)
    if rank == 0:
        value = args.steps * S_STEPS * world / dt
        workload = '%s: %d events x %d seg x 500-d C3D (%s), S=%d decoder timesteps, V1=%d, %s, one video per GPU' % (
            'c3' if args.mode == 'train' else 'c2', N_EV, A_SEG, 'T_v=160 overlapping' if args.overlap else 'disjoint rows, T_v=8192',
            S_STEPS, V1, 'fwd+bwd+clamp+Adam' if args.mode == 'train' else 'forward + loss only (train-mode dropout)')
        if args.c5:
            workload = ('c5: SST (2-layer LSTM 500->512, K=256) over one 256-segment video + caption path on %d proposals of 4..256 '
                        'segments, S=%d, V1=%d, joint fwd+bwd+clamp+Adam, one video per GPU' % (N_EV, S_STEPS, V1))
        out = {
            'metric': 'caption-decoder timesteps/sec (fwd+bwd)' if args.mode == 'train' else 'caption-decoder timesteps/sec (fwd only)', 'value': round(value, 1), 'unit': 'timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (storage and accumulation are fp32; the reverse recurrences use native fp32 MFMAs; the 11 large batched projections '
                     'and the forward recurrences run as fp16-pair "h2" MFMA emulation: 2 scaled fp16 planes per operand, 3 products, fp32 '
                     'accumulate, error <= native fp32 MFMA vs float64; native_f32 = the same run with every product on fp32 MFMAs)',
            'data': 'synthetic',
            'config': {'workload': workload,
                       'global_events': N_EV * world, 'timesteps_per_step': S_STEPS, 'parallelism': 'dp%d' % world,
                       'final_loss': round(loss, 5),
                       'host_path': host_path},
        }
        out['config'].update(dp_info)
        if native is not None:
            out['native_f32'] = native
        if roof is not None:
            out['roofline'] = roof
        if world == 1 and not args.no_cpu and args.mode == 'train':
            out['cpu_baseline'] = cpu_leg(args)
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
