"""The path bench.py TIMES, at the configurations it is timed at, held directly to the reference's fixtures and to the oracle (-m gpu).

bench.py's headline iteration is `FusedTrainStep` (echr_train_step: host-side targets / masks -> active-row compaction, fused criterion,
three-stream backward tail, clamp + Adam inside the call) on the `c3bench` layout; its other lines are the forward-only form of the same
entry (BASELINE config 2) and the joint 'tap_cg' form (`prepare` + `tap_grad` + `defer_update`, BASELINE config 5).  The tests of
tests/test_gpu_parity.py pin the autograd path at those sizes and the one-call path on small shapes; these pin the one-call path itself:
  reference protocol: train.py:281-317 (zero_grad, forward, criterion, backward, clip_gradient, Adam), misc/utils.py:66-75, :107-111.
"""
import numpy as np
import pytest
import torch

from echr_amd import synth
from oracle import summary as SM
from tests import util as U
from tests.test_gpu_parity import TOL_GRAD, TOL_LOSS

pytestmark = pytest.mark.gpu


def _fused(opt, params, train_mode=True, lr=None, clip=None):
    from echr_amd.fused import FusedTrainStep
    from echr_amd.optim import ClampAdam
    m = U.build_gpu_model(opt, params, train_mode)
    o = ClampAdam(m.parameters(), lr=opt.lr if lr is None else lr, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon,
                  arena=m.build_arena())
    return m, o, FusedTrainStep(m, o, grad_clip=clip)


def _device_inputs(vid):
    dev = torch.device('cuda')
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    labels = torch.from_numpy(vid['labels'])
    # exactly what bench.py hands the call (bench.py: tgt_h, msk_h): numpy views of labels[:, 1:] / masks[:, 1:] on the HOST
    return tap, c3d, lda, labels, labels[:, 1:].numpy(), vid['masks'][:, 1:]


def _check_grad_summaries(g, mode, grads, tag='|grad|'):
    for key, v in SM.summarize_grads(grads).items():
        ref = g[mode + tag + key]
        name = key.split('|')[0]
        if name in U.NOISE_ONLY:
            assert np.abs(np.asarray(v)).max() < 1e-6 and np.abs(np.asarray(ref)).max() < 1e-6, (key, v, ref)
            continue
        scale = max(float(g[mode + tag + name + '|linf']), U.GRAD_FLOOR)
        if key.endswith('|l2') or key.endswith('|linf'):
            assert abs(float(v) - float(ref)) < TOL_GRAD * max(abs(float(ref)), U.GRAD_FLOOR), (key, float(v), float(ref))
        else:
            assert np.abs(v - ref).max() < TOL_GRAD * scale, (key, np.abs(v - ref).max() / scale)


@pytest.mark.parametrize('case', ['c3bench', 'c2full', 'c2'])
def test_timed_path_loss_and_gradients_vs_reference_fixture_and_oracle(case):
    """(a) echr_train_step with step=False at the timed layout: the loss and the gradient summaries against the REFERENCE's own outputs
    (case_<name>.npz, train mode with the injected dropout masks), then every gradient element against the oracle."""
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    m, o, f = _fused(opt, params)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
    torch.cuda.synchronize()
    assert 0 < f.last_active_rows < tgt_h.size                    # the compacted (active-row) form ran, as in the timed region
    assert abs(loss - float(g['train|loss'])) < TOL_LOSS * abs(float(g['train|loss'])), (loss, float(g['train|loss']))
    grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None) for k, p in m.named_parameters()}
    _check_grad_summaries(g, 'train', grads)
    _, rloss, rgrads = U.run_oracle(opt, params, vid, True)
    assert abs(loss - rloss) < TOL_LOSS * abs(rloss)
    for k, rg in rgrads.items():
        if rg is None:
            assert grads[k] is None, k
        else:
            assert U.grad_close(k, grads[k], rg, TOL_GRAD), (k, U.relerr(grads[k], rg))


@pytest.mark.parametrize('case', ['c3bench', 'c2full'])
def test_timed_path_one_optimizer_step_vs_oracle(case):
    """(b) ONE full iteration exactly as bench.py issues it (step=True: clamp + Adam inside the call, grad_clip and lr of the recipe):
    parameters and Adam moments against the oracle's clamp_adam_step applied to the ORACLE's gradients (misc/utils.py:107-111 +
    torch.optim.Adam, train.py:315-317)."""
    from oracle import echr_ref_cpu as O
    opt, params, vid = synth.make_case(case)
    m, o, f = _fused(opt, params, clip=opt.grad_clip)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h))
    torch.cuda.synchronize()
    _, rloss, rgrads = U.run_oracle(opt, params, vid, True)
    assert abs(loss - rloss) < TOL_LOSS * abs(rloss)
    assert o._flat['step'] == 1
    ar = m._echr_arena
    lr = opt.lr
    for i, (k, p) in enumerate(m.named_parameters()):
        assert ar.params[i] is p
        lo, n = ar.offsets[i], p.numel()
        mom = o._flat['m'][lo:lo + n].view(p.shape).cpu().numpy()
        var = o._flat['v'][lo:lo + n].view(p.shape).cpu().numpy()
        new = p.detach().cpu().numpy()
        if rgrads[k] is None:             # never-used parameters: untouched, like torch.optim.Adam skipping grad None
            assert np.array_equal(new, params[k]) and not mom.any() and not var.any(), k
            continue
        rp, rm, rv = (torch.from_numpy(x.copy()) for x in (params[k], np.zeros_like(params[k]), np.zeros_like(params[k])))
        O.clamp_adam_step(rp, torch.from_numpy(rgrads[k]), rm, rv, 1, lr, opt.optim_alpha, opt.optim_beta, opt.optim_epsilon, opt.grad_clip)
        if k in U.NOISE_ONLY:             # the true gradient is exactly zero: the update is a coin flip of +-lr on both sides
            assert np.abs(new - params[k]).max() <= 1.01 * lr
            continue
        assert U.grad_close(k, mom, rm.numpy(), TOL_GRAD), (k, 'exp_avg', U.relerr(mom, rm.numpy()))
        gmax = float(np.abs(rgrads[k]).max())
        assert np.abs(var - rv.numpy()).max() <= 2.5 * TOL_GRAD * max(float(rv.max()), 1e-3 * U.GRAD_FLOOR ** 2) + 1e-20, (k, 'exp_avg_sq')
        dgpu, dref = new - params[k], rp.numpy() - params[k]
        assert np.abs(dgpu).max() <= 1.01 * lr and np.abs(dgpu - dref).max() <= 2.01 * lr, k
        # Adam's first update is lr * g / (|g| + eps): where |g| is resolvable against the noise of its tensor the two sides agree closely
        solid = np.abs(rgrads[k]) > 1e-4 * gmax
        if solid.any():
            assert np.abs(dgpu - dref)[solid].max() < 0.02 * lr, (k, np.abs(dgpu - dref)[solid].max() / lr)


@pytest.mark.parametrize('case', ['c3bench', 'c2full'])
@pytest.mark.parametrize('train_mode', [True, False])
def test_timed_path_forward_only_loss_vs_reference_fixture(case, train_mode):
    """(c) bench.py --mode fwd (BASELINE config 2): forward + criterion through the same one-call entry, against the reference's loss
    (train mode: the injected dropout masks; eval mode: dropout off)."""
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    m, o, f = _fused(opt, params, train_mode)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, forward_only=True))
    ref = float(g[('train' if train_mode else 'eval') + '|loss'])
    assert abs(loss - ref) < TOL_LOSS * abs(ref), (loss, ref)
    assert o._flat is None or o._flat['step'] == 0
    assert all(p.grad is None for p in m.parameters())


@pytest.mark.parametrize('train_mode', [True, False])
def test_timed_joint_iteration_vs_reference_fixture(train_mode):
    """(d) bench.py --c5 (BASELINE config 5, train.py:300-313): `prepare()` before the proposal encoder's forward, the caption side as one
    call with `tap_grad` + `defer_update` + `prepared`, then `autograd.backward([tap_loss, tap_feats], [None, tap_grad])` into the SST --
    both losses, the caption model's gradients and the SST's gradients against the reference's own (case_c5.npz).  The call steps the
    optimiser; with no clip and the gradients read back from the arena behind the join, the update does not disturb what is compared."""
    from echr_amd import models as EM
    from echr_amd.misc.utils import TAPModelCriterion
    opt, params, sst_params, vid = synth.make_c5()
    g = U.gold('case_c5.npz')
    mode = 'train' if train_mode else 'eval'
    m, o, f = _fused(opt, params, train_mode, lr=1e-9)
    dev = torch.device('cuda')
    tapm = EM.setup_tap(opt)
    tapm.load_state_dict({k: torch.from_numpy(v) for k, v in sst_params.items()})
    tapm = tapm.to(dev)
    tapm.eval()                                          # the fixture's SST runs without inter-layer dropout
    _, c3d, lda, labels, tgt_h, msk_h = _device_inputs(dict(vid, tap=np.zeros((1, 1), np.float32)))
    tl, tm, tw = (torch.from_numpy(vid[k]) for k in ('tap_labels', 'tap_masks', 'w1'))
    f.prepare(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
    tap_feats, props = tapm(c3d)
    g_tap = torch.zeros_like(tap_feats)
    tap_loss = TAPModelCriterion()(props, tm, tl, tw)
    cg_loss = f(tap_feats.detach(), c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tap_grad=g_tap, defer_update=True, prepared=True)
    assert opt.lambda2 == 1.0                            # (the call backpropagates 1 * cg_loss; the fixture's joint loss is lambda1 * tap + lambda2 * cg)
    torch.autograd.backward([opt.lambda1 * tap_loss, tap_feats], [None, g_tap])
    f.join()
    torch.cuda.synchronize()
    assert abs(float(tap_loss) - float(g[mode + '|tap_loss'])) < TOL_LOSS * abs(float(g[mode + '|tap_loss']))
    assert abs(float(cg_loss) - float(g[mode + '|cg_loss'])) < TOL_LOSS * abs(float(g[mode + '|cg_loss'])), (float(cg_loss), float(g[mode + '|cg_loss']))
    assert np.abs(tap_feats.detach().cpu().numpy()[::8, ::16] - g['tap_feats|slice']).max() < 1e-5
    ar = m._echr_arena
    unused = {'lm_model.core.fusion_layer.weight', 'lm_model.core.fusion_layer.bias', 'fusion_model.h2a_layer.weight', 'fusion_model.h2a_layer.bias'}
    grads = {}
    for i, (k, p) in enumerate(m.named_parameters()):
        gv = ar.grad_view(i).detach().cpu().numpy()
        if k in unused:
            assert not gv.any(), k
            grads[k] = None
        else:
            grads[k] = gv
    _check_grad_summaries(g, mode, grads)
    _check_grad_summaries(g, mode, {k: p.grad.detach().cpu().numpy() for k, p in tapm.named_parameters()}, tag='|sstgrad|')


def test_prepare_keeps_device_criterion_inputs_alive():
    """`prepare()` with DEVICE targets / masks: the converted (contiguous) copies are only referenced by raw pointers in the argument
    struct; allocations the caller makes between `prepare()` and the `prepared=True` call must not be able to recycle them."""
    opt, params, vid = synth.make_case('c2')
    dev = torch.device('cuda')
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    tgt_d, msk_d = torch.from_numpy(np.ascontiguousarray(tgt_h)).to(dev), torch.from_numpy(np.ascontiguousarray(msk_h)).to(dev)

    def run(device_inputs):
        m, o, f = _fused(opt, params, lr=1e-9)
        a = (tgt_d, msk_d) if device_inputs else (tgt_h, msk_h)
        f.prepare(c3d, lda, labels, vid['ind'], vid['soi'], *a)
        junk = [torch.full((tgt_d.numel(),), 7, device=dev, dtype=tgt_d.dtype) for _ in range(8)]        # same sizes as the temporaries
        junk += [torch.full((msk_d.numel(),), 3.0, device=dev) for _ in range(8)]
        g_tap = torch.zeros_like(tap)
        loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], *a, tap_grad=g_tap, prepared=True))
        f.join()
        torch.cuda.synchronize()
        del junk
        return loss, m._echr_arena.flat_g.clone()

    lh, gh = run(False)
    ld, gd = run(True)
    assert abs(lh - ld) < 2e-6 * abs(lh), (lh, ld)
    assert float((gh - gd).abs().max()) <= 2e-5 * float(gh.abs().max())


# ---- training parity in the PEAKED regime (synth.PEAKED, tests/golden/case_peaked.npz) --------------------------------------------------
# Logits reach +-200 and log-probs -370 there (fp32 ulp 3e-5): two CORRECT fp32 evaluations differ by ~1e-4 in a log-prob and by 4-5e-6 of
# a gradient tensor's max-norm (tests/test_oracle_golden.py::test_peaked_regime_noise_floor_of_fp32_itself: the reference's arithmetic on 8
# threads vs 1 thread; against float64 1.25e-4 / 5e-6).  Gates: log-probs 1e-3 absolute (= 3e-6 of their magnitude, 10x that noise, 30x
# inside north_star's 1e-4 relative); loss TOL_LOSS; gradients TOL_GRAD (1e-5) on the default (h2) path -- i.e. at 2x the reference's own
# noise -- and 3e-5 for the alternative product paths.
TOL_LOGP_PEAKED = 1e-3
TOL_GRAD_PEAKED_ALT = 3e-5

def _peaked():
    g = U.gold('case_peaked.npz')
    opt, params, vid = synth.make_peaked(g['labels'], g['masks'])
    return g, opt, params, vid


@pytest.mark.parametrize('train_mode', [True, False])
def test_peaked_regime_full_path_vs_oracle_and_reference(train_mode):
    """Forward + criterion + backward where the softmax is PEAKED (top-1 probability > 0.9 on 70 % of the active rows in train mode, one
    target in ten confidently wrong): d logits = softmax - onehot then spans > 2^40 inside a row and inside a 256-wide k segment of the
    fp16-pair ("h2") operand format of the default configuration.  Every log-prob and every gradient element against the oracle; loss and
    summaries against the reference's own outputs (OldModel_NEW.py:136, misc/utils.py:66-75)."""
    g, opt, params, vid = _peaked()
    mode = 'train' if train_mode else 'eval'
    pred, loss, grads, _ = U.run_gpu(opt, params, vid, train_mode)
    rpred, rloss, rgrads = U.run_oracle(opt, params, vid, train_mode)
    if train_mode:
        act = vid['masks'][:, 1:1 + pred.shape[1]] > 0
        assert (np.exp(rpred.max(2))[act] > 0.9).mean() > 0.6
    assert np.abs(pred - rpred).max() < TOL_LOGP_PEAKED, np.abs(pred - rpred).max()
    assert abs(loss - rloss) < TOL_LOSS * abs(rloss) and abs(loss - float(g[mode + '|loss'])) < TOL_LOSS * abs(float(g[mode + '|loss']))
    s = SM.summarize_logp(pred)
    safe = g[mode + '|logp|margin'] > 1e-3
    assert np.array_equal(s['argmax'][safe], g[mode + '|logp|argmax'][safe])
    for k, rg in rgrads.items():
        if rg is None:
            assert grads[k] is None, k
        else:
            assert U.grad_close(k, grads[k], rg, TOL_GRAD), (k, U.relerr(grads[k], rg))
    _check_grad_summaries(g, mode, grads)


def test_peaked_regime_timed_path_vs_reference():
    """The same regime through echr_train_step (the path bench.py times: active-row compaction, criterion fused into the logits pass, h2
    operands packed from the compacted d logits): loss and gradients against the reference's fixture and the oracle, then one full step."""
    g, opt, params, vid = _peaked()
    m, o, f = _fused(opt, params)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
    torch.cuda.synchronize()
    assert 0 < f.last_active_rows < tgt_h.size
    assert abs(loss - float(g['train|loss'])) < TOL_LOSS * abs(float(g['train|loss'])), (loss, float(g['train|loss']))
    grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None) for k, p in m.named_parameters()}
    _check_grad_summaries(g, 'train', grads)
    _, rloss, rgrads = U.run_oracle(opt, params, vid, True)
    for k, rg in rgrads.items():
        if rg is not None:
            assert U.grad_close(k, grads[k], rg, TOL_GRAD), (k, U.relerr(grads[k], rg))
    # the three product paths agree in this regime too (h2 default above; exact three-plane bf16 split; native fp32 MFMAs)
    from echr_amd import _lib
    lib = _lib.load()
    try:
        for cfg in ((0, 1), (0, 0)):
            lib.echr_config_set(b'gemm_h2', cfg[0]); lib.echr_config_set(b'gemm_bf16x3', cfg[1]); lib.echr_config_set(b'persist_h2', cfg[0])
            m.set_dropout_state(U.SEED, U.OFFSET)
            l2 = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
            torch.cuda.synchronize()
            assert abs(l2 - rloss) < TOL_LOSS * abs(rloss), (cfg, l2, rloss)
            for k, p in m.named_parameters():
                if rgrads[k] is not None:
                    assert U.grad_close(k, p.grad.detach().cpu().numpy(), rgrads[k], TOL_GRAD_PEAKED_ALT), (cfg, k, U.relerr(p.grad.detach().cpu().numpy(), rgrads[k]))
    finally:
        lib.echr_config_set(b'gemm_h2', 1); lib.echr_config_set(b'gemm_bf16x3', 1); lib.echr_config_set(b'persist_h2', 1)


# ---- scene-context variants (CaptionGenerator.py:87-104: 'VC' = c3d_feats.mean(0), 'VH' = tap_feats.mean(0)) ------------------------------

@pytest.mark.parametrize('vt', ['VLVCVH', 'VC', 'VLVH'])
def test_scene_context_variants_vs_oracle(vt):
    """Every combination's scene vector feeds stream 2's gates: log-probs, loss, every parameter gradient AND d tap_feats (through 'VH' and
    through the event encoder's anchors) against the oracle; 'VLVCVH' is additionally pinned to the reference by case_vctx.npz."""
    from echr_amd.misc.utils import LanguageModelCriterion
    from oracle import echr_ref_cpu as O
    from tests.test_gpu_parity import TOL_LOGP
    opt = synth.default_opt(vocab_size=300, seq_length=7, video_context_type=vt)
    params = synth.make_params(opt, 4)
    vid = synth.make_video(12, 40, 9, 301, seed=61)
    m = U.build_gpu_model(opt, params, True)
    dev = torch.device('cuda')
    tap = torch.from_numpy(vid['tap']).to(dev).requires_grad_(True)
    c3d, lda = torch.from_numpy(vid['c3d']).to(dev), torch.from_numpy(vid['lda']).to(dev)
    labels, masks = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    pred = m(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train')
    loss = LanguageModelCriterion()(pred, labels[:, 1:].to(dev), masks[:, 1:].to(dev))
    loss.backward()
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
    otap = torch.from_numpy(vid['tap']).requires_grad_(True)
    opred = O.caption_forward(P, otap, torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), labels, vid['ind'], vid['soi'], 'train',
                              U.oracle_drop(opt), opt.n_head, video_context_type=vt)
    oloss = O.lm_criterion(opred, labels[:, 1:], masks[:, 1:])
    oloss.backward()
    assert np.abs(pred.detach().cpu().numpy() - opred.detach().numpy()).max() < TOL_LOGP
    assert abs(float(loss) - float(oloss)) < TOL_LOSS * abs(float(oloss))
    for k, p in m.named_parameters():
        if P[k].grad is None:
            assert p.grad is None, k
        else:
            assert U.grad_close(k, p.grad.cpu().numpy(), P[k].grad.numpy(), TOL_GRAD), (k, U.relerr(p.grad.cpu().numpy(), P[k].grad.numpy()))
    assert U.grad_close('tap_feats', tap.grad.cpu().numpy(), otap.grad.numpy(), TOL_GRAD), U.relerr(tap.grad.cpu().numpy(), otap.grad.numpy())
    if 'VH' in vt:          # every row of tap_feats receives the mean's share of the scene-context gradient
        assert float((tap.grad.abs().sum(1) > 0).float().mean()) == 1.0


def test_scene_context_through_the_one_call_path():
    """echr_train_step with a 'VLVC' scene vector (formed ahead of the call): loss and gradients against the oracle; 'VH' with d tap_feats
    asked for: the scene vector's gradient comes back in `tap_grad`."""
    opt = synth.default_opt(vocab_size=300, seq_length=7, video_context_type='VLVC')
    params = synth.make_params(opt, 4)
    vid = synth.make_video(12, 40, 9, 301, seed=61)
    _, rloss, rgrads = U.run_oracle(opt, params, vid, True)
    m, o, f = _fused(opt, params)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
    assert abs(loss - rloss) < TOL_LOSS * abs(rloss)
    for k, p in m.named_parameters():
        if rgrads[k] is not None:
            assert U.grad_close(k, p.grad.cpu().numpy(), rgrads[k], TOL_GRAD), (k, U.relerr(p.grad.cpu().numpy(), rgrads[k]))
    # 'VH' alone, d tap_feats asked for (round 6: routed by the library, echr_train_step_args.vh_offset): against the oracle with tap as a leaf;
    # prepare() runs before tap_feats exist and still declines
    from oracle import echr_ref_cpu as O
    opt2 = synth.default_opt(vocab_size=300, seq_length=7, video_context_type='VH')
    p2 = synth.make_params(opt2, 4)
    m2, o2, f2 = _fused(opt2, p2)
    g_tap = torch.zeros_like(tap)
    l2 = float(f2(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False, tap_grad=g_tap))
    P = {k: torch.from_numpy(v.copy()) for k, v in p2.items()}
    tap_c = torch.from_numpy(vid['tap'].copy()).requires_grad_(True)
    lab, msk = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    pred = O.caption_forward(P, tap_c, torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), lab, vid['ind'], vid['soi'], 'train', U.oracle_drop(opt2),
                             opt2.n_head, video_context_type='VH', event_context_type=opt2.event_context_type, fST_type='fST0', use_posit=opt2.use_posit,
                             init_feats_type=opt2.CG_init_feats_type)
    rl = O.lm_criterion(pred, lab[:, 1:], msk[:, 1:])
    rl.backward()
    assert abs(l2 - float(rl)) < TOL_LOSS * abs(float(rl))
    assert U.grad_close('tap_feats', g_tap.cpu().numpy(), tap_c.grad.numpy(), TOL_GRAD), U.relerr(g_tap.cpu().numpy(), tap_c.grad.numpy())
    with pytest.raises(NotImplementedError):
        f2.prepare(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)


# ---- the event encoder's other gate / affinity combinators (MA_attention_8_NEW.py:148-157) and use_posit = 0 ------------------------------

@pytest.mark.parametrize('train_mode', [True, False])
@pytest.mark.parametrize('case', ['fst1', 'fst2', 'fst3', 'noposit'])
def test_event_encoder_combinators_vs_oracle_and_reference(case, train_mode):
    """fST1 gate + aff, fST2 log(clamp(gate)) + aff, fST3 the gate alone, use_posit = 0 the affinity alone: log-probs, loss, every gradient against
    the oracle, loss / summaries / greedy sequence against the reference's fixture.  Parameters the chosen combination does not reach (query / key
    under fST3, the pair MLP without the position branch) have `grad None` in the reference; here their gradient is None or exactly zero (the
    flat-arena optimiser treats both alike: no moment, no update)."""
    from tests.test_gpu_parity import TOL_LOGP
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    mode = 'train' if train_mode else 'eval'
    pred, loss, grads, m = U.run_gpu(opt, params, vid, train_mode)
    rpred, rloss, rgrads = U.run_oracle(opt, params, vid, train_mode)
    assert np.abs(pred - rpred).max() < TOL_LOGP
    assert abs(loss - rloss) < TOL_LOSS * abs(rloss) and abs(loss - float(g[mode + '|loss'])) < TOL_LOSS * abs(float(g[mode + '|loss']))
    unreached = [k for k, v in rgrads.items() if v is None and 'fusion_layer' not in k and 'h2a_layer' not in k]
    assert len(unreached) == (0 if case in ('fst1', 'fst2') else 4), unreached
    for k, rg in rgrads.items():
        if rg is None:
            assert grads[k] is None or not np.any(grads[k]), k
        else:
            assert U.grad_close(k, grads[k], rg, TOL_GRAD), (k, U.relerr(grads[k], rg))
    # (fST1 / fST3: the fc2 bias shifts every softmax input of a row alike, so its true gradient is exactly zero -- both sides hold rounding noise
    # ~1e-10, compared on the absolute scale by grad_close above, not by the relative summary gate)
    noise = ('fusion_model.enc_attn.pair_pos_fc2.bias',) if case in ('fst1', 'fst3') else ()
    _check_grad_summaries(g, mode, {k: v for k, v in grads.items() if rgrads[k] is not None and k not in noise})
    if not train_mode:
        dev = torch.device('cuda')
        with torch.no_grad():
            seq, _ = m(torch.from_numpy(vid['tap']).to(dev), torch.from_numpy(vid['c3d']).to(dev), torch.from_numpy(vid['lda']).to(dev), [], vid['ind'], vid['soi'], mode='eval')
        assert np.array_equal(seq.cpu().numpy(), g['sample|seq'])


def test_event_encoder_combinators_through_the_one_call_path():
    """echr_train_step with fST2 and without the position branch: loss and gradients against the oracle."""
    for case in ('fst2', 'noposit'):
        opt, params, vid = synth.make_case(case)
        _, rloss, rgrads = U.run_oracle(opt, params, vid, True)
        m, o, f = _fused(opt, params)
        tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
        loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
        assert abs(loss - rloss) < TOL_LOSS * abs(rloss), (case, loss, rloss)
        for k, p in m.named_parameters():
            if rgrads[k] is not None:
                assert U.grad_close(k, p.grad.cpu().numpy(), rgrads[k], TOL_GRAD), (case, k, U.relerr(p.grad.cpu().numpy(), rgrads[k]))
            else:
                assert p.grad is None or not bool(p.grad.abs().max() > 0), (case, k)


@pytest.mark.parametrize('case', ['init', 'initc'])
def test_initial_state_map_vs_oracle_and_module_contract(case):
    """CG_init_feats_type (OldModel_NEW.py:72-96): `init_hidden` returns the SAME [3,N,H] tensor for h and c -- init_linear over
    cat([scene | event | clip.mean(1)]), the clip mean over the PADDED frame slots -- element-wise against the oracle, from the zero-copy clip
    view and from the padded [N,A,D] tensor an external caller passes; the one-call step refuses the option (the recipe has a zero state)."""
    from echr_amd import functional as EF
    from oracle import echr_ref_cpu as O
    opt, params, vid = synth.make_case(case)
    m = U.build_gpu_model(opt, params, False)
    dev = torch.device('cuda')
    tap, c3d, lda = (torch.from_numpy(vid[k]).to(dev) for k in ('tap', 'c3d', 'lda'))
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    with torch.no_grad():
        ovideo = O.video_context(torch.from_numpy(vid['lda']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['tap']), opt.video_context_type)
        oevent = O.event_context(P, torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), vid['ind'], vid['soi'], opt.n_head, None)
        oclip, _ = O.clip_context(torch.from_numpy(vid['c3d']), vid['soi'])
        oh, oc = O.init_hidden(P, ovideo, oevent, oclip, opt.CG_init_feats_type)
        ev = EF.event_index_tensors(vid['soi'], vid['ind'], dev)
        video = m.get_video_context(tap, c3d, lda, vid['ind'], vid['soi'])
        event = oevent.to(dev)
        view, _ = m.get_clip_context(tap, c3d, lda, vid['ind'], vid['soi'], _ev=ev)
        clip, clip_mask = m.get_clip_context(tap, c3d, lda, vid['ind'], vid['soi'])
        h_view, c_view = m.lm_model.init_hidden(video, event, view)
        h_pad, _ = m.lm_model.init_hidden(video, event, clip, clip_mask)
    assert h_view is c_view and tuple(h_view.shape) == (3, len(vid['soi']), opt.CG_rnn_size)
    assert U.relerr(h_view.cpu().numpy(), oh.numpy()) < 2e-6
    assert U.relerr(h_pad.cpu().numpy(), oh.numpy()) < 2e-6
    _fused(opt, params)          # (round 6: the one-call path takes the initial state too -- test_one_call_path_option_variants_* hold it to the fixtures)


def test_handover_callback_points_and_error_propagation():
    """echr_train_step_args.handover_cb: with step=False and handover=True the library calls back on the host once per hand-over point (0 = logit
    layer, 1 = LSTM layers), from inside the call, with the library stream that carries the range's last writer; what the callback queues on that
    stream runs behind the point (here: a copy of the range, which must equal the final gradient).  An exception raised inside the callback
    cannot cross the C frame: it is kept and re-raised behind the call."""
    opt, params, vid = synth.make_case('c2')
    m, o, f = _fused(opt, params, clip=opt.grad_clip)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    ar = m._echr_arena
    lm = m.lm_model
    slots = sorted(ar.slot(p) for p in (lm.logit.weight, lm.logit.bias))
    lo, hi = ar.span(slots)
    seen, copies = [], {}

    def cb(which, stream_ptr):
        seen.append(which)
        if which == 0:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr, device=ar.flat_g.device)):
                copies[0] = ar.flat_g[lo:hi].clone()          # queued on the library's stream: behind the logit layer's last gradient launch

    f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False, handover=True, handover_cb=cb)
    torch.cuda.synchronize()
    assert sorted(seen) == [0, 1]
    assert torch.equal(copies[0], ar.flat_g[lo:hi])              # the range was final at the point
    assert float(copies[0].abs().max()) > 0

    def bad(which, stream_ptr):
        raise RuntimeError('boom %d' % which)
    with pytest.raises(RuntimeError, match='boom'):
        f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False, handover=True, handover_cb=bad)
    torch.cuda.synchronize()
    # the object stays usable
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False))
    assert np.isfinite(loss)


# ---- optimiser step counts against updates the device skipped (asynchronous -62) -------------------------------------------------------
def _abort_some_iterations(lib, issue, n_calls=3):
    """Issue iterations while ONE hand-off wait of the forward recurrence never completes (persist_inject_timeout): every persistent launch
    queued meanwhile drains through its bounded spins, the optimiser kernels behind it skip.  The -62 may surface at ANY of the sites that
    check (the next call, prepare(), ClampAdam.step, the explicit check below) -- wherever it does, it goes through _lib.check."""
    from echr_amd import _lib
    n_err = 0
    try:
        assert lib.echr_config_set(b'persist_spin_limit', 20000) == 0
        assert lib.echr_config_set(b'persist_inject_timeout', 200000 + 3) == 0
        for _ in range(n_calls):
            try:
                issue()
            except _lib.EchrHipError as e:
                assert 'rc=-62' in str(e), e
                n_err += 1
        torch.cuda.synchronize()
    finally:
        lib.echr_config_set(b'persist_inject_timeout', 0)
        lib.echr_config_set(b'persist_spin_limit', 0)
    try:
        _lib.check(lib.echr_check_async(), 'drain')
    except _lib.EchrHipError:
        n_err += 1
    assert n_err >= 1
    assert lib.echr_check_async() == 0


def _assert_counts_consistent(o):
    applied = int(o._applied.item())
    assert o._issued == applied, (o._issued, applied)
    if o._flat is not None:
        assert o._flat['step'] == applied, (o._flat['step'], applied)
    return applied


def test_step_count_after_async_abort_one_call_path():
    """after an asynchronous -62 the optimiser's step count must equal the number of updates the device APPLIED (Adam's bias
    correction depends on it), the parameters and moments must be those of the applied updates, and training must continue as if the lost
    iterations had never been issued -- checked against a twin that runs the same number of clean iterations (eval-mode dropout)."""
    from echr_amd import _lib
    lib = _lib.load()
    opt, params, vid = synth.make_case('c2')
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    args = (tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
    m, o, f = _fused(opt, params, train_mode=False, lr=1e-3, clip=opt.grad_clip)
    f(*args)
    torch.cuda.synchronize()
    assert _assert_counts_consistent(o) == 1
    snap = (m._echr_arena.flat_p.clone(), o._flat['m'].clone(), o._flat['v'].clone())
    _abort_some_iterations(lib, lambda: f(*args))
    assert _assert_counts_consistent(o) == 1                     # every update queued behind the aborted launches was skipped, and un-counted
    assert torch.equal(m._echr_arena.flat_p, snap[0]) and torch.equal(o._flat['m'], snap[1]) and torch.equal(o._flat['v'], snap[2])
    f(*args)
    torch.cuda.synchronize()
    assert _assert_counts_consistent(o) == 2
    m2, o2, f2 = _fused(opt, params, train_mode=False, lr=1e-3, clip=opt.grad_clip)
    f2(*args); f2(*args)
    torch.cuda.synchronize()
    dp = (m._echr_arena.flat_p - m2._echr_arena.flat_p).abs()
    # Adam's first updates move every parameter by ~lr whatever its gradient's size, so elements whose gradient sits at the noise floor of the
    # split-K atomics differ by up to 2 lr between ANY two runs; everything else agrees closely.  A step count of 3 instead of 2 would scale
    # the WHOLE second update by (1 - 0.9^2) / (1 - 0.9^3) * sqrt((1 - 0.999^3) / (1 - 0.999^2)) = 0.86: 0.14 lr on every element
    assert float(dp.max()) <= 2.01e-3
    assert float((dp > 0.05e-3).float().mean()) < 0.02, float((dp > 0.05e-3).float().mean())
    assert float((o._flat['m'] - o2._flat['m']).abs().max()) <= 1e-4 * float(o2._flat['m'].abs().max())


def test_step_count_after_async_abort_two_optimisers():
    """The joint 'tap_cg' iteration as bench.py --c5 issues it: prepare() + the deferred caption update + the proposal encoder's own
    ClampAdam.  Both optimisers queue updates behind an aborted launch; each winds ITS count back by ITS skipped updates (the advisor's case:
    a process-wide counter charged both skips to the caption optimiser and none to the other)."""
    from echr_amd import _lib
    from echr_amd import models as EM
    from echr_amd.misc.utils import TAPModelCriterion, clip_gradient
    from echr_amd.optim import ClampAdam
    lib = _lib.load()
    opt, params, sst_params, vid = synth.make_c5()
    m, o, f = _fused(opt, params, train_mode=False, lr=1e-4, clip=opt.grad_clip)
    dev = torch.device('cuda')
    tapm = EM.setup_tap(opt)
    tapm.load_state_dict({k: torch.from_numpy(v) for k, v in sst_params.items()})
    tapm = tapm.to(dev)
    tapm.eval()
    tap_ar = tapm.build_arena()
    tap_o = ClampAdam(tapm.parameters(), lr=1e-4, arena=tap_ar)
    _, c3d, lda, labels, tgt_h, msk_h = _device_inputs(dict(vid, tap=np.zeros((1, 1), np.float32)))
    tl, tm, tw = (torch.from_numpy(vid[k]).to(dev) for k in ('tap_labels', 'tap_masks', 'w1'))
    crit = TAPModelCriterion()

    def iteration():
        tap_o.zero_grad()
        try:
            f.prepare(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
            tap_feats, props = tapm(c3d)
            g_tap = torch.zeros_like(tap_feats)
            tap_loss = 0.01 * crit(props, tm, tl, tw)
            f(tap_feats.detach(), c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tap_grad=g_tap, defer_update=True, prepared=True)
            torch.autograd.backward([tap_loss, tap_feats], [None, g_tap])
            clip_gradient(tap_o, opt.grad_clip)
            tap_o.step()
        except _lib.EchrHipError:
            f.cancel_prepare()
            raise

    iteration()
    f.join()
    torch.cuda.synchronize()
    assert _assert_counts_consistent(o) == 1 and _assert_counts_consistent(tap_o) == 1
    snap = (m._echr_arena.flat_p.clone(), tap_ar.flat_p.clone())
    _abort_some_iterations(lib, iteration)
    f.join()
    torch.cuda.synchronize()
    a_cg, a_tap = _assert_counts_consistent(o), _assert_counts_consistent(tap_o)
    # the caption update sits behind the aborted caption recurrence in every injected iteration; the proposal encoder's update of such an
    # iteration is queued behind it too.  Whatever WAS applied, count and state agree per optimiser:
    assert a_cg == 1, a_cg
    assert torch.equal(m._echr_arena.flat_p, snap[0])
    if a_tap == 1:
        assert torch.equal(tap_ar.flat_p, snap[1])
    iteration()
    f.join()
    torch.cuda.synchronize()
    assert _assert_counts_consistent(o) == a_cg + 1 and _assert_counts_consistent(tap_o) == a_tap + 1


# ---- the reference's non-recipe options on the ONE-CALL path (round 6) ------------------------------------------------------------------
@pytest.mark.parametrize('case', ['er1', 'er2', 'init', 'initc', 'vctx'])
@pytest.mark.parametrize('train_mode', [True, False])
def test_one_call_path_option_variants_vs_reference_fixture_and_oracle(case, train_mode):
    """echr_train_step with event_context_type 'ER1' / 'ER2' (CaptionGenerator.py:106-130), a non-zero initial decoder state
    (CG_init_feats_type 'VEC' / 'C', OldModel_NEW.py:72-96) and the 'VL' + 'VC' + 'VH' scene context (CaptionGenerator.py:87-104): the loss and
    the gradient summaries against the REFERENCE's own outputs (case_<name>.npz), every gradient element and d loss / d tap_feats (the joint
    iteration's `tap_grad`, which for 'VH' also carries the scene vector's gradient) against the oracle."""
    from oracle import echr_ref_cpu as O
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    mode = 'train' if train_mode else 'eval'
    m, o, f = _fused(opt, params, train_mode)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    g_tap = torch.zeros_like(tap)
    loss = float(f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, step=False, tap_grad=g_tap))
    torch.cuda.synchronize()
    assert abs(loss - float(g[mode + '|loss'])) < TOL_LOSS * abs(float(g[mode + '|loss'])), (loss, float(g[mode + '|loss']))
    grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else None) for k, p in m.named_parameters()}
    _check_grad_summaries(g, mode, {k: v for k, v in grads.items() if v is not None})
    # the oracle, with tap_feats as a leaf
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
    tap_c = torch.from_numpy(vid['tap'].copy()).requires_grad_(True)
    lab, msk = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    pred = O.caption_forward(P, tap_c, torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), lab, vid['ind'], vid['soi'], 'train',
                             U.oracle_drop(opt) if train_mode else None, opt.n_head, video_context_type=opt.video_context_type,
                             event_context_type=opt.event_context_type, fST_type=getattr(opt, 'fST_type', 'fST0'), use_posit=opt.use_posit,
                             init_feats_type=opt.CG_init_feats_type)
    rloss = O.lm_criterion(pred, lab[:, 1:], msk[:, 1:])
    rloss.backward()
    assert abs(loss - float(rloss)) < TOL_LOSS * abs(float(rloss))
    for k, p in P.items():
        if p.grad is None or not bool(p.grad.any()):
            assert grads[k] is None or not grads[k].any() or U.grad_close(k, grads[k], np.zeros_like(grads[k]), TOL_GRAD), k
        else:
            assert grads[k] is not None and U.grad_close(k, grads[k], p.grad.numpy(), TOL_GRAD), (k, U.relerr(grads[k], p.grad.numpy()))
    if tap_c.grad is None or not bool(tap_c.grad.any()):          # 'ER1' without 'VH': tap_feats do not reach the loss
        assert not bool(g_tap.any())
    else:
        assert U.grad_close('tap_feats', g_tap.cpu().numpy(), tap_c.grad.numpy(), TOL_GRAD), U.relerr(g_tap.cpu().numpy(), tap_c.grad.numpy())


@pytest.mark.parametrize('case', ['init', 'vctx'])
def test_one_call_path_option_variants_full_step_matches_autograd_path(case):
    """... and as a full iteration (clamp + Adam inside the call, `defer_update` asked for and declined where the options need the whole call):
    parameters after one step against the autograd path's on a twin model."""
    from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
    from echr_amd.optim import ClampAdam
    opt, params, vid = synth.make_case(case)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    m, o, f = _fused(opt, params, False, lr=1e-3, clip=opt.grad_clip)
    f(tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tap_grad=torch.zeros_like(tap), defer_update=True)
    f.join()
    m2 = U.build_gpu_model(opt, params, False)
    o2 = ClampAdam(m2.parameters(), lr=1e-3, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=m2.build_arena())
    dev = torch.device('cuda')
    crit = LanguageModelCriterion()
    o2.zero_grad()
    crit(m2(tap, c3d, lda, labels, vid['ind'], vid['soi'], mode='train'), labels[:, 1:].to(dev), torch.from_numpy(vid['masks'])[:, 1:].to(dev)).backward()
    clip_gradient(o2, opt.grad_clip)
    o2.step()
    torch.cuda.synchronize()
    for (k, p), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        d = (p - p2).abs()
        assert float(d.max()) <= 2.01e-3, k                                   # (Adam's first step: +-lr per element)
        if k not in U.NOISE_ONLY:
            assert float((d > 0.05e-3).float().mean()) < 0.02, (k, float((d > 0.05e-3).float().mean()))


@pytest.mark.parametrize('train_mode', [True, False])
@pytest.mark.parametrize('early,order', [(True, 'after'), (False, 'after'), (True, 'hook'), (False, 'hook')])
def test_joint_train_step_vs_reference_fixture(train_mode, early, order):
    """fused.JointTrainStep (what bench.py --c5 times since round 6): the proposal encoder run through the library directly, its backward
    and update issued from echr_train_step's mid-call hook right behind d tap_feats, the caption side's tail behind that -- both losses, the
    caption model's gradients and the SST's gradients against the reference's own joint iteration (case_c5.npz; train.py:292-329)."""
    from echr_amd import models as EM
    from echr_amd.fused import JointTrainStep
    from echr_amd.optim import ClampAdam
    opt, params, sst_params, vid = synth.make_c5()
    g = U.gold('case_c5.npz')
    mode = 'train' if train_mode else 'eval'
    m, o, f = _fused(opt, params, train_mode, lr=1e-9)
    dev = torch.device('cuda')
    tapm = EM.setup_tap(opt)
    tapm.load_state_dict({k: torch.from_numpy(v) for k, v in sst_params.items()})
    tapm = tapm.to(dev)
    tapm.eval()                                          # the fixture's SST runs without inter-layer dropout
    tap_ar = tapm.build_arena()
    tap_o = ClampAdam(tapm.parameters(), lr=1e-9, arena=tap_ar)
    j = JointTrainStep(f, tapm, tap_o, lambda1=opt.lambda1, early_prepare=early, order=order)
    _, c3d, lda, labels, tgt_h, msk_h = _device_inputs(dict(vid, tap=np.zeros((1, 1), np.float32)))
    tl, tm, tw = (torch.from_numpy(vid[k]).to(dev) for k in ('tap_labels', 'tap_masks', 'w1'))
    assert opt.lambda2 == 1.0
    for rep in range(2):          # (twice: the second call re-uses every buffer and joins the first one's deferred tail)
        total = j(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tm, tl, tw)
        assert f.mid_called == (order == 'hook')         # the deferred form ran and called the hook
        f.join()
        torch.cuda.synchronize()
        tap_loss, cg_loss = float(j.tap_loss), float(j.cg_loss)
        assert abs(tap_loss - float(g[mode + '|tap_loss'])) < TOL_LOSS * abs(float(g[mode + '|tap_loss']))
        assert abs(cg_loss - float(g[mode + '|cg_loss'])) < TOL_LOSS * abs(float(g[mode + '|cg_loss'])), (cg_loss, float(g[mode + '|cg_loss']))
        assert abs(float(total) - (opt.lambda1 * tap_loss + cg_loss)) < 1e-5 * abs(float(total))
        assert np.abs(j._bufs['tap'].cpu().numpy()[::8, ::16] - g['tap_feats|slice']).max() < 1e-5
        ar = m._echr_arena
        unused = {'lm_model.core.fusion_layer.weight', 'lm_model.core.fusion_layer.bias', 'fusion_model.h2a_layer.weight', 'fusion_model.h2a_layer.bias'}
        grads = {}
        for i, (k, p) in enumerate(m.named_parameters()):
            gv = ar.grad_view(i).detach().cpu().numpy()
            if k in unused:
                assert not gv.any(), k
            else:
                grads[k] = gv
        _check_grad_summaries(g, mode, grads)
        _check_grad_summaries(g, mode, {k: tap_ar.grad_view(i).detach().cpu().numpy() for i, (k, p) in enumerate(tapm.named_parameters())}, tag='|sstgrad|')
        assert o._flat['step'] == rep + 1 and tap_o._flat['step'] == rep + 1
        assert int(o._applied.item()) == rep + 1 and int(tap_o._applied.item()) == rep + 1
        if train_mode:
            m.set_dropout_state(U.SEED, U.OFFSET)        # the fixture's dropout masks again for the second pass


def test_stage_ahead_with_changing_videos():
    """Stage-ahead (csrc/step.hip): from the second consecutive `echr_train_step` on, the index staging copy and the event encoder's position
    embedding of iteration i + 1 run on the tail stream BESIDE iteration i's clamp + Adam.  Six back-to-back iterations (no host sync in
    between) on videos whose event lists, lengths and captions all differ -- a staging copy or an embedding that ran too early or too late
    would feed one iteration the other's indices -- against the autograd path stepping a twin model through the same videos
    (train.py:281-317 six times)."""
    from echr_amd.misc.utils import LanguageModelCriterion, clip_gradient
    from echr_amd.optim import ClampAdam
    opt, params, _ = synth.make_case('c2')
    dev = torch.device('cuda')
    vids = [synth.make_video(64 if i % 2 == 0 else 48, 128 if i % 3 else 96, 21, opt.CG_vocab_size + 1, seed=900 + i, T_v=160 + 16 * i,
                             video_dim=opt.video_dim, hidden_dim=opt.hidden_dim, lda_dim=opt.lda_dim) for i in range(6)]
    m, o, f = _fused(opt, params, False, lr=1e-3, clip=opt.grad_clip)
    m2 = U.build_gpu_model(opt, params, False)
    o2 = ClampAdam(m2.parameters(), lr=1e-3, betas=(opt.optim_alpha, opt.optim_beta), eps=opt.optim_epsilon, arena=m2.build_arena())
    crit = LanguageModelCriterion()
    pinned = [tuple(torch.from_numpy(v[k]).pin_memory() for k in ('tap', 'c3d', 'lda')) for v in vids]
    losses, dev_in = [], []
    for v, pin in zip(vids, pinned):          # back to back: the host runs ahead, every iteration but the first stages ahead
        # this iteration's features arrive by an ASYNCHRONOUS upload queued on the caller's stream right in front of the call: everything of
        # the call that reads them -- on whichever library stream -- has to be ordered behind the caller's stream's position at entry
        tap, c3d, lda = (x.to(dev, non_blocking=True) for x in pin)
        dev_in.append((tap, c3d, lda))
        labels = torch.from_numpy(v['labels'])
        losses.append(f(tap, c3d, lda, labels, v['ind'], v['soi'], labels[:, 1:].numpy(), v['masks'][:, 1:]))
    torch.cuda.synchronize()
    ref = []
    for v, (tap, c3d, lda) in zip(vids, dev_in):
        labels = torch.from_numpy(v['labels'])
        o2.zero_grad()
        loss = crit(m2(tap, c3d, lda, labels, v['ind'], v['soi'], mode='train'), labels[:, 1:].to(dev), torch.from_numpy(v['masks'])[:, 1:].to(dev))
        loss.backward()
        clip_gradient(o2, opt.grad_clip)
        o2.step()
        ref.append(float(loss))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(losses, ref)):
        # (the trajectories separate slowly: Adam moves noise-floor gradient elements by +-lr on both sides independently)
        assert abs(float(a) - b) < 2e-3 * abs(b), (i, float(a), b)
    assert abs(float(losses[0]) - ref[0]) < 1e-5 * abs(ref[0])
    assert o._flat['step'] == 6 and int(o._applied.item()) == 6


# ---- fixed-order accumulation (round 6: echr_config_set("deterministic", 1)) ------------------------------------------------------------
def _bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize('case', ['c2', 'vctx', 'initc', 'er2'])
def test_deterministic_switch_two_runs_agree_bit_for_bit(case):
    """`echr_amd.set_deterministic(True)` (echr_config_set("deterministic", 1)): two runs of FusedTrainStep from the same parameters, inputs and
    dropout seed give the same BITS -- the loss, every parameter gradient, d loss / d tap_feats, and after three full iterations (clamp + Adam
    inside the call) the parameters and both moment vectors.  The reference's CPU path has this property at a fixed thread count; the default
    configuration (split-K atomics, persistent recurrences with atomic exchange adds) does not -- tools/det_probe.py lists what differs there.
    Overlapping events (shared d P_all rows), repeated tokens (the embedding scatter) and repeated anchor rows are all present in these cases."""
    import echr_amd
    opt, params, vid = synth.make_case(case)
    tap, c3d, lda, labels, tgt_h, msk_h = _device_inputs(vid)
    args = (tap, c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h)
    assert len(set(np.asarray(vid['ind']).tolist())) < len(vid['ind']) or case != 'c2'          # c2 repeats anchor rows

    def run():
        torch.manual_seed(99)
        m, o, f = _fused(opt, params, True, lr=1e-3, clip=opt.grad_clip)
        g_tap = torch.zeros_like(tap)
        loss = f(*args, step=False, tap_grad=g_tap)
        torch.cuda.synchronize()
        out = {'loss': _bits(loss.reshape(1)), 'tap_grad': _bits(g_tap)}
        out.update({'g|' + k: _bits(p.grad) for k, p in m.named_parameters() if p.grad is not None})
        losses = [f(*args) for _ in range(3)]
        torch.cuda.synchronize()
        out['losses'] = _bits(torch.stack([x.reshape(()) for x in losses]))
        out['p'], out['m'], out['v'] = _bits(m._echr_arena.flat_p), _bits(o._flat['m']), _bits(o._flat['v'])
        assert o._flat['step'] == 3 and int(o._applied.item()) == 3
        return out

    echr_amd.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        echr_amd.set_deterministic(False)
    assert a.keys() == b.keys() and len(a) > 30
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()), a[k].size)


def test_deterministic_switch_joint_iteration_bit_for_bit():
    """... and the joint 'tap_cg' iteration (fused.JointTrainStep: proposal encoder forward / criterion / backward + both updates): both losses and
    both models' parameters after two iterations, bit for bit between two runs."""
    import echr_amd
    from echr_amd import models as EM
    from echr_amd.fused import JointTrainStep
    from echr_amd.optim import ClampAdam
    opt, params, sst_params, vid = synth.make_c5()
    dev = torch.device('cuda')
    _, c3d, lda, labels, tgt_h, msk_h = _device_inputs(dict(vid, tap=np.zeros((1, 1), np.float32)))
    tl, tm, tw = (torch.from_numpy(vid[k]).to(dev) for k in ('tap_labels', 'tap_masks', 'w1'))

    def run():
        torch.manual_seed(7)
        m, o, f = _fused(opt, params, True, lr=1e-3, clip=opt.grad_clip)
        tapm = EM.setup_tap(opt)
        tapm.load_state_dict({k: torch.from_numpy(v) for k, v in sst_params.items()})
        tapm = tapm.to(dev)
        tapm.eval()
        tap_o = ClampAdam(tapm.parameters(), lr=1e-3, arena=tapm.build_arena())
        j = JointTrainStep(f, tapm, tap_o, lambda1=opt.lambda1, tap_grad_clip=opt.grad_clip)
        out = {}
        for it in range(2):
            total = j(c3d, lda, labels, vid['ind'], vid['soi'], tgt_h, msk_h, tm, tl, tw)
            f.join()
            torch.cuda.synchronize()
            out['loss%d' % it] = _bits(torch.stack([total.reshape(()), j.tap_loss.reshape(()), j.cg_loss.reshape(())]))
        out['p'], out['tap_p'] = _bits(m._echr_arena.flat_p), _bits(tapm._echr_arena.flat_p)
        return out

    echr_amd.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        echr_amd.set_deterministic(False)
    for k in a:
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()), a[k].size)
