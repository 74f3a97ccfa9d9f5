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


class Workload(object):
    """One timed configuration: model, optimiser, inputs and the closure that issues ONE iteration of it.
    kind 'train' = BASELINE config 3 (the headline), 'fwd' = config 2 (forward + criterion), 'c5' = config 5 (SST + caption path, joint)."""

    def __init__(self, args, kind, rank, dev, use_dist):
        import echr_amd
        from echr_amd import parallel
        from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
        from echr_amd.optim import ClampAdam
        self.kind, self.args, self.use_dist = kind, args, use_dist
        c5 = kind == 'c5'
        opt, params, vid = make_workload(rank, args.overlap, c5)
        self.opt, self.vid = opt, vid
        model = echr_amd.CaptionGenerator(opt)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        self.model = model = model.to(dev).train()
        crit = LanguageModelCriterion()
        arena = None if args.no_arena else model.build_arena()      # flat parameter/gradient buffers: 1-launch Adam, 1-bucket all-reduce
        optim = ClampAdam(model.parameters(), lr=opt.lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=arena)
        tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
        labels = torch.from_numpy(vid['labels'])                       # host copy: step count needs no device sync
        tgt = labels[:, 1:].to(dev)
        msk = torch.from_numpy(vid['masks'])[:, 1:].to(dev)
        tgt_h, msk_h = labels[:, 1:].numpy(), vid['masks'][:, 1:]
        ind, soi = vid['ind'], vid['soi']

        # the whole iteration as ONE library call (echr_train_step: include/echr_hip.h, echr_amd/fused.py) instead of ~75 ctypes calls and four
        # autograd nodes -- for EVERY world size: with several ranks the call stops behind the backward pass (step=False) and hands the
        # gradient ranges that are final early to the collective stream while its tail still runs (fused.DataParallelStep)
        fused = dp = None
        if args.fused != 'off' and arena is not None and not (c5 and use_dist):
            from echr_amd.fused import FusedTrainStep
            fused = FusedTrainStep(model, optim, grad_clip=opt.grad_clip)
            if use_dist and kind == 'train':
                from echr_amd.fused import DataParallelStep
                dp = DataParallelStep(fused, overlap=os.environ.get('ECHR_DP_OVERLAP', '1') != '0')
        elif use_dist and arena is not None and os.environ.get('ECHR_DP_OVERLAP', '1') != '0':
            parallel.enable_overlap(model)       # autograd path: LSTM-layer gradients are reduced while the backward tail runs
        self.fused, self.dp = fused, dp
        self.active_rows = None

        if c5:
            from echr_amd import models as EM
            from echr_amd.misc.utils import TAPModelCriterion
            torch.manual_seed(0)
            tap_model = EM.setup_tap(opt).to(dev)
            tap_model.train()
            tap_arena = None if args.no_arena else tap_model.build_arena()
            tap_optim = ClampAdam(tap_model.parameters(), lr=opt.lr, arena=tap_arena)
            tap_crit = TAPModelCriterion()
            tl, tm, tw = (torch.from_numpy(vid[k]).to(dev) for k in ('tap_labels', 'tap_masks', 'w1'))

            joint = None
            if fused is not None and tap_arena is not None and os.environ.get('ECHR_JOINT_STEP', '1') != '0':
                # round 6: the proposal side without an autograd graph, its backward + update issued from inside the caption call right behind
                # d tap_feats (fused.JointTrainStep, echr_train_step_args.mid_cb).  ECHR_JOINT_STEP=0: the round-5 form below (A/B)
                from echr_amd.fused import JointTrainStep
                joint = JointTrainStep(fused, tap_model, tap_optim, lambda1=0.01, tap_grad_clip=opt.grad_clip,
                                       early_prepare=os.environ.get('ECHR_EARLY_PREPARE', '1') != '0')

            def iteration():      # train.py's joint 'tap_cg' iteration: SST -> caption path -> lambda1*tap_loss + lambda2*cg_loss
                if joint is not None:
                    return joint(c3d, lda, labels, ind, soi, tgt_h, msk_h, tm, tl, tw)
                if fused is not None:
                    # the caption side as ONE library call (forward, criterion, backward, clip, Adam); d loss / d tap_feats comes back in g_tap and
                    # is backpropagated into the proposal encoder together with its own loss (same sums as loss.backward() of the joint loss)
                    tap_optim.zero_grad()
                    early = os.environ.get('ECHR_EARLY_PREPARE', '1') != '0'
                    if early:          # the caption side's tap-independent part starts now and runs beside the proposal encoder's forward
                        fused.prepare(c3d, lda, labels, ind, soi, tgt_h, msk_h)
                    tap_feats, props = tap_model(c3d)
                    g_tap = torch.zeros_like(tap_feats)
                    tap_loss = 0.01 * tap_crit(props, tm, tl, tw)          # (queued before the caption side: nothing of it waits for g_tap)
                    defer = os.environ.get('ECHR_DEFER_UPDATE', '1') != '0'
                    cg_loss = fused(tap_feats.detach(), c3d, lda, labels, ind, soi, tgt_h, msk_h, tap_grad=g_tap, defer_update=defer, prepared=early)
                    torch.autograd.backward([tap_loss, tap_feats], [None, g_tap])
                    clip_gradient(tap_optim, opt.grad_clip)
                    tap_optim.step()
                    return tap_loss.detach() + cg_loss
                optim.zero_grad()
                tap_optim.zero_grad()
                tap_feats, props = tap_model(c3d)
                pred = model(tap_feats, c3d, lda, labels, ind, soi, mode='train')
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
        elif kind == 'fwd':
            def iteration():
                if fused is not None:          # forward + criterion only (BASELINE config 2), same one-call entry
                    return fused(tap, c3d, lda, labels, ind, soi, tgt_h, msk_h, forward_only=True)
                with torch.no_grad():
                    return crit(model(tap, c3d, lda, labels, ind, soi, mode='train'), tgt, msk)
        else:
            def iteration():
                if dp is not None:
                    return dp(tap, c3d, lda, labels, ind, soi, tgt_h, msk_h)
                if fused is not None:
                    # (criterion inputs handed over on the host, as the reference's loader produces them: they travel with the index vectors)
                    return fused(tap, c3d, lda, labels, ind, soi, tgt_h, msk_h)
                optim.zero_grad()
                pred = model(tap, c3d, lda, labels, ind, soi, mode='train')
                loss = crit(pred, tgt, msk)
                loss.backward()
                if use_dist:
                    parallel.allreduce_gradients(model, force=True)        # SUM over ranks == reference m_batch accumulation
                clip_gradient(optim, opt.grad_clip)
                optim.step()
                return loss
        self.iteration = iteration

    def host_path(self):
        if self.dp is not None:
            return 'echr_train_step + early range collectives'
        return 'echr_train_step (one library call per iteration)' if self.fused is not None else 'autograd Functions (one ctypes call per fused region)'

    def describe(self, args):
        if self.kind == 'c5':
            return ('c5: SST (2-layer LSTM 500->512, K=256) over one 256-segment video + caption path on %d proposals of 4..256 '
                    'segments, S=%d, V1=%d, joint fwd+bwd+clamp+Adam, one video per GPU' % (N_EV, S_STEPS, V1))
        return '%s: %d events x %d seg x 500-d C3D (%s), S=%d decoder timesteps, V1=%d, %s, one video per GPU' % (
            'c3' if self.kind == 'train' else 'c2', N_EV, A_SEG, 'T_v=160 overlapping' if args.overlap else 'disjoint rows, T_v=8192',
            S_STEPS, V1, 'fwd+bwd+clamp+Adam' if self.kind == 'train' else 'forward + loss only (train-mode dropout)')

    def active_rows_str(self):
        """Label positions that can reach the loss (up to each caption's last non-zero mask entry) of all N x S: what the late-fusion and
        weight-gradient products of a training iteration run on; the CPU baseline computes all rows."""
        mk = self.vid['masks'][:, 1:1 + S_STEPS] != 0
        live = np.flip(np.logical_or.accumulate(np.flip(mk, 1), 1), 1)
        return '%d/%d' % (int(live.sum()), mk.size)


def timed_regions(iteration, steps, regions, fence, world, dev, use_dist):
    """`regions` back-to-back timed regions of exactly `steps` iterations, each bracketed by barrier + torch.cuda.synchronize() on both sides
    and reduced with MAX over ranks; returns the per-region wall times [s], the host issue time of the first region and the last loss."""
    out, t_issue, loss = [], None, None
    import gc
    gc.collect()
    if os.environ.get('ECHR_BENCH_GC') != '1':
        gc.disable()          # like timeit: a generation-2 collection of the interpreter (~40 ms with torch loaded) is not part of an iteration
    try:
        for r in range(regions):
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = iteration()
            if t_issue is None:
                t_issue = time.perf_counter() - t0          # host time to issue the timed steps (the GPU may still be working)
            fence()
            dt = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            out.append(dt)
    finally:
        gc.enable()          # (an exception inside iteration() must not leave the collector off for the legs that follow)
    return out, t_issue, loss


def region_stats(times, steps, world):
    ms = sorted(1e3 * t / steps for t in times)
    med = float(np.median(ms))
    return dict(ms_per_step=round(med, 3), value=round(1e3 * S_STEPS * world / med, 1),
                regions=dict(n=len(ms), steps_each=steps, ms_per_step_min=round(ms[0], 3), ms_per_step_median=round(med, 3), ms_per_step_max=round(ms[-1], 3)))


# kernel classes of libechr_hip.so (echr_prof_read kinds) -> (name, rocprof kernel symbol, bound)
PROF_KINDS = {0: ('gemm_f32_kernel', 'gemm_f32_kernel<*> (all tile/layout instantiations)', 'mfma'),
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


def roofline_pass(wl, args, rank, fence, brief=False, variant=''):
    """Instrumented pass over the same iterations: HIP events around every launch of each kernel class, on the stream the kernel is launched
    on.  EVERY rank runs the iterations (they contain the gradient collectives); only rank 0 instruments and reports."""
    from echr_amd import _lib
    lib = _lib.load()
    n_it = max(2, min(args.steps, 5))
    if rank == 0:
        lib.echr_prof_enable(1)
    for _ in range(n_it):
        wl.iteration()
    fence()
    if rank != 0:
        return None
    stats = {}
    for k, (name, sym, bound) in PROF_KINDS.items():
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        lib.echr_prof_read(k, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n))
        stats[name] = dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value, sym=sym, bound=bound)
    lib.echr_prof_enable(0)
    # HBM traffic per launch cannot be collected inside a timed run (PMC needs rocprofv3 --pmc in separate passes): it is read from the
    # committed summary of the SAME command (tools/pmc_traffic.sh -> profiles/rNN_pmc_traffic*.json); null when absent
    # (one pair of files per workload: `_c5` for config 5; other variants -- overlapping rows, forward only -- carry no counters)
    # (`variant` = '_native': the native_f32 configuration of the same workload has counter files of its own)
    sfx = ('_c5' if wl.kind == 'c5' else '') + variant
    traffic, traffic_src, mfma_pmc = {}, None, {}
    if not (args.overlap or wl.kind == 'fwd'):
        for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
            name = '%s_pmc_traffic%s.json' % (rnd, sfx)
            try:
                traffic = json.load(open(os.path.join(ROOT, 'profiles', name)))
                traffic_src = 'profiles/' + name
                break
            except Exception:
                pass
        # MFMA activity (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD-cycles of the dispatch) likewise comes from the committed --pmc pass of the
        # same command (tools/pmc_mfma.sh -> profiles/rNN_pmc_mfma*.json)
        for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
            try:
                mfma_pmc = json.load(open(os.path.join(ROOT, 'profiles', '%s_pmc_mfma%s.json' % (rnd, sfx))))
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
        out = dict(bound='mfma' if st['bound'] == 'valu' else st['bound'], kernel=st['sym'] if not brief else name, achieved=round(ach, 2), peak=peak, unit=unit,
                   frac=round(ach / peak, 4), traffic=tb)
        if brief:
            out.update(avg_launch_us=round(1e3 * ms / st['launches'], 2), launches_per_step=st['launches'] / n_it, ms_per_step=round(ms / n_it, 3))
            return out
        out.update(traffic_source=traffic_src if tb else None, traffic_commit=traffic_commit if tb else None,
                   algorithmic_bytes_per_launch=round(alg) if alg else None,
                   traffic_over_algorithmic=round(tb / alg, 2) if (tb and alg) else None,
                   mfma_busy_frac_pmc=mfma_busy(name),
                   avg_launch_us=round(1e3 * ms / st['launches'], 2), event_overhead_us_subtracted=round(ev_us, 2),
                   launches_per_step=st['launches'] / n_it, ms_per_step=round(ms / n_it, 3))
        return out

    dom = max(stats, key=lambda k: stats[k]['ms'])
    roof = line(dom)
    # `achieved` = algorithmic flops (2MNK) or bytes per launch / HIP-event duration on the launch stream; `peak` for the
    # MFMA-bound kernels: 157.3 TF (native fp32 MFMA: gemm_f32, rec_gemm), 2500/3 TF (gemm_h2: three fp16 MFMA products per
    # fp32-grade product), 2500/6 TF (gemm_split: six bf16 products).
    if not brief:
        roof['other_kernels'] = {k: line(k) for k in stats if k != dom and line(k) is not None}
    return roof


def gpu_leg(args, rank, world, local_rank):
    from echr_amd import _lib
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if os.environ.get('ECHR_STREAMS_FIRST', '1') != '0':
        # the library's helper streams first (a no-op if main() already did it ahead of the RCCL communicator): their hardware queues do not
        # depend on what else in the process creates streams before the first iteration
        _lib.check(_lib.load().echr_streams_init(), 'streams_init')
    use_dist = dist.is_available() and dist.is_initialized()
    kind = 'c5' if args.c5 else ('fwd' if args.mode == 'fwd' else 'train')
    wl = Workload(args, kind, rank, dev, use_dist)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def say(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    say('model ready; warm-up')
    for i in range(args.warmup):
        wl.iteration()
        torch.cuda.synchronize()
        say('warm-up %d done' % i)
    times, t_issue, loss = timed_regions(wl.iteration, args.steps, args.regions, fence, world, dev, use_dist)
    head = region_stats(times, args.steps, world)
    say('host issue time %.3f ms per step; timed regions (ms per step): %s' % (1e3 * t_issue / args.steps, ' '.join('%.3f' % (1e3 * t / args.steps) for t in times)))
    if os.environ.get('ECHR_HOST_PROBE') == '1':          # diagnostic: host time of one iteration issued into an empty queue
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t1 = time.perf_counter(); wl.iteration(); ts.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        say('host time of one iteration issued into an empty queue: min %.3f ms, median %.3f ms' % (1e3 * min(ts), 1e3 * sorted(ts)[5]))
    final_loss = float(loss.item())

    # second timed figure with every large projection on the NATIVE fp32 MFMA path (no fp16-pair / bf16-plane emulation)
    native = None
    if not args.no_native and kind == 'train':
        lib = _lib.load()
        for k in (b'gemm_h2', b'persist_h2', b'gemm_bf16x3'):
            lib.echr_config_set(k, 0)
        for _ in range(2):
            wl.iteration()
        tn, _, _ = timed_regions(wl.iteration, args.steps, min(args.regions, 3), fence, world, dev, use_dist)
        st = region_stats(tn, args.steps, world)
        native = dict(value=st['value'], unit='timesteps/s', ms_per_step=st['ms_per_step'], regions=st['regions'],
                      note='same workload with gemm_h2=0, gemm_bf16x3=0, persist_h2=0: every product on v_mfma_f32_* (exact fp32 MFMA)')
        if not args.no_roofline:
            # the same instrumented pass as the headline's, in THIS configuration: its dominant class is the fp32-MFMA products
            native['roofline'] = roofline_pass(wl, args, rank, fence, variant='_native')
        for k in (b'gemm_h2', b'persist_h2', b'gemm_bf16x3'):
            lib.echr_config_set(k, 1)
        for _ in range(2):
            wl.iteration()
        fence()

    roof = None
    if not args.no_roofline:
        roof = roofline_pass(wl, args, rank, fence)

    # several ranks: how much of the gradient exchange the backward tail did NOT hide (HIP events around the caller's stream's wait for the
    # last collective, fused.DataParallelStep), per step, in a pass of its own behind the timed regions; MAX over ranks like the timing
    exchange = None
    if wl.dp is not None and use_dist:
        wl.dp.measure = True
        for _ in range(max(3, min(args.steps, 10))):
            wl.iteration()
        fence()
        wl.dp.measure = False
        exchange = wl.dp.exchange_report()
        t = torch.tensor([exchange['exposed_ms_median'] or 0.0, exchange['exposed_ms_max'] or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange['exposed_ms_median'], exchange['exposed_ms_max'] = round(float(t[0]), 4), round(float(t[1]), 4)
        exchange['note'] = ('time the caller\'s stream waits for the collectives behind the backward pass (max over ranks); the early ranges '
                            'start inside the backward tail, the remainder behind it')

    # the other single-GPU configurations of BASELINE.json in the same process: config 2 (forward + criterion only, same one-call entry) and
    # config 5 (SST proposal encoder + caption path, joint iteration) -- each with its own timed regions and its dominant kernel's roofline
    others = None
    if kind == 'train' and world == 1 and not args.no_others and not args.overlap:
        others = {}
        for name, k2 in (('c2', 'fwd'), ('c5', 'c5')):
            w2 = Workload(args, k2, rank, dev, use_dist)
            for _ in range(max(3, args.warmup // 2)):
                w2.iteration()
            t2, _, l2 = timed_regions(w2.iteration, args.steps, args.regions, fence, world, dev, use_dist)
            st = region_stats(t2, args.steps, world)
            rf = None if args.no_roofline else roofline_pass(w2, args, rank, fence, brief=True)
            others[name] = dict(metric='caption-decoder timesteps/sec (%s)' % ('fwd only' if k2 == 'fwd' else 'SST + decoder, joint fwd+bwd'),
                                value=st['value'], unit='timesteps/s', ms_per_step=st['ms_per_step'], regions=st['regions'],
                                workload=w2.describe(args), final_loss=round(float(l2.item()), 5), host_path=w2.host_path(), roofline=rf)
            if hasattr(w2.fused, 'join'):
                w2.fused.join()
            torch.cuda.synchronize()
            del w2
            say('%s: %.3f ms per step' % (name, st['ms_per_step']))
    return dict(head=head, loss=final_loss, roof=roof, native=native, exchange=exchange, host_path=wl.host_path(), workload=wl.describe(args),
                active_rows=wl.active_rows_str() if kind != 'fwd' else None, others=others)


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
                    help='one echr_train_step call per iteration (auto = on, for every world size) or the autograd path (off)')
    ap.add_argument('--regions', type=int, default=5, help='back-to-back timed regions of --steps iterations each; ms_per_step = their median')
    ap.add_argument('--no-others', action='store_true', help='skip the config-2 / config-5 lines measured after the headline (other_configs)')
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
            # the library's helper streams BEFORE RCCL's: the runtime hands out its few hardware queues in stream-creation order, and helper
            # streams created behind the communicator's share queues with each other (the backward tail's three streams then serialise:
            # 1.64 vs 1.49 ms per iteration on one MI355X with a single-rank communicator, tools/dp_host_profile.py)
            if os.environ.get('ECHR_STREAMS_FIRST', '1') != '0':
                from echr_amd import _lib
                _lib.check(_lib.load().echr_streams_init(), 'streams_init')
            # RCCL prints a version banner to STDOUT when the communicator comes up; this program's stdout is ONE JSON line
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
                dist.barrier()
            finally:
                os.dup2(keep, 1)
                os.close(keep)
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
        # One rank per GPU: plain persistent launches.  Every collective of an iteration is queued behind the reverse recurrence (hand-over
        # callbacks / the end of the backward pass) and the caller's stream waits for the last of them before clamp + Adam, i.e. before the next
        # iteration's forward recurrence -- by stream order no collective kernel is ever resident beside a persistent pair, and should one be,
        # the pair's bounded waits end in -ETIME, not in a hang.  The cooperative launch (a grid starts only when all of its workgroups can be
        # resident) costs 0.8 ms per iteration once RCCL is up in the process (tools/dp_host_profile.py: 2.67 vs 1.83 ms) and is kept for
        # what needs it: the one-GPU rehearsal, where two ranks' grids share a device (ECHR_PERSIST_COOP overrides).
        from echr_amd import _lib
        coop = int(os.environ.get('ECHR_PERSIST_COOP', '1' if os.environ.get('ECHR_BENCH_ONE_GPU') == '1' else '0'))
        _lib.load().echr_config_set(b'persist_coop', coop)
        dp_info.update(persist_coop=coop, dp_algo=__import__('echr_amd.parallel', fromlist=['choose_algo']).choose_algo(world),
                       dp_overlap=os.environ.get('ECHR_DP_OVERLAP', '1') != '0', dp_via=os.environ.get('ECHR_DP_VIA', 'callback'),
                       streams_first=os.environ.get('ECHR_STREAMS_FIRST', '1') != '0')
        if os.environ.get('ECHR_BENCH_ONE_GPU') == '1' and os.environ.get('ECHR_BENCH_PERSIST', '0') == '0':
            # rehearsal with every rank on ONE device: two 256-workgroup persistent grids must not share it -> launch-per-phase recurrences
            _lib.load().echr_config_set(b'persist', 0)
            _lib.load().echr_config_set(b'persist_bwd', 0)
    res = gpu_leg(args, rank, world, local_rank)
    if rank == 0:
        head = res['head']
        out = {
            'metric': 'caption-decoder timesteps/sec (fwd+bwd)' if args.mode == 'train' else 'caption-decoder timesteps/sec (fwd only)', 'value': head['value'], 'unit': 'timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (storage and accumulation are fp32; the reverse recurrences use native fp32 MFMAs; the 11 large batched projections '
                     'and the forward recurrences run as fp16-pair "h2" MFMA emulation: 2 block-scaled fp16 planes per operand, 3 products, fp32 '
                     'accumulate; gradients within 1e-5 of the reference on its random-init AND peaked-softmax fixtures (tests/golden/case_peaked.npz); '
                     'native_f32 = the same run with every product on fp32 MFMAs)',
            'data': 'synthetic',
            'config': {'workload': res['workload'],
                       'global_events': N_EV * world, 'timesteps_per_step': S_STEPS, 'parallelism': 'dp%d' % world,
                       'final_loss': round(res['loss'], 5),
                       'host_path': res['host_path'],
                       'timed_regions': head['regions']},
        }
        if res['active_rows']:
            out['config']['active_rows'] = res['active_rows']
        out['config'].update(dp_info)
        if res.get('exchange'):
            out['config']['exchange'] = res['exchange']
            out['config']['exchange_exposed_ms'] = res['exchange']['exposed_ms_median']
        if res['native'] is not None:
            out['native_f32'] = res['native']
            out['config']['native_f32'] = {k: res['native'][k] for k in ('value', 'ms_per_step')}
        if res['others']:
            # (also under `config`: a consumer that keeps only the contract's keys still sees them)
            out['other_configs'] = res['others']
            out['config']['other_configs'] = res['others']
        if res['roof'] is not None:
            out['roofline'] = res['roof']
        if world == 1 and not args.no_cpu and args.mode == 'train':
            out['cpu_baseline'] = cpu_leg(args)
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
