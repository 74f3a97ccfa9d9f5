"""CPU: the oracle against the fixtures the reference itself produced (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from echr_amd import synth
from oracle import echr_ref_cpu as O
from oracle import summary as SM
from tests import util as U


def test_position_matrix_and_embedding():
    g = U.gold('position.npz')
    for k in ('a', 'b'):
        pm = O.position_matrix(g[k + '|soi'])
        assert np.array_equal(pm, g[k + '|pos_matrix'])
        assert np.array_equal(O.position_embedding(pm, 512).astype(np.float32), g[k + '|pos_emb_f32'])
        assert np.array_equal(O.position_embedding(pm, 32).astype(np.float32), g[k + '|pos_emb32_f32'])


@pytest.mark.parametrize('train_mode', [False, True])
def test_tiny_full_tensors(train_mode):
    opt, params, vid = synth.make_case('tiny')
    g = U.gold('case_tiny.npz')
    mode = 'train' if train_mode else 'eval'
    pred, loss, grads = U.run_oracle(opt, params, vid, train_mode)
    assert np.abs(pred - g[mode + '|logp']).max() < 1e-6
    assert abs(loss - float(g[mode + '|loss'])) < 1e-6
    for k, v in grads.items():
        if v is None:
            assert (mode + '|grad|' + k) not in g          # the two never-used parameter groups
        else:
            assert U.relerr(v, g[mode + '|grad|' + k]) < 1e-5, k


@pytest.mark.parametrize('case', ['c1', 'c2', 'c3bench', 'vctx', 'er1', 'er2', 'fst1', 'fst2', 'fst3', 'noposit', 'init', 'initc'])
def test_config_summaries(case):
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    pred, loss, grads = U.run_oracle(opt, params, vid, True)
    assert abs(loss - float(g['train|loss'])) < 1e-5
    s = SM.summarize_logp(pred)
    assert np.abs(s['slice'] - g['train|logp|slice']).max() < 1e-5
    assert np.array_equal(s['argmax'], g['train|logp|argmax'])
    for key, v in SM.summarize_grads(grads).items():
        ref = g['train|grad|' + key]
        assert np.allclose(v, ref, rtol=1e-4, atol=1e-7 * (1 + np.abs(ref).max())), key


def test_peaked_regime_summaries():
    """The oracle on the peaked-softmax training case (synth.PEAKED; captions from the fixture) against the reference's own summaries."""
    g = U.gold('case_peaked.npz')
    opt, params, vid = synth.make_peaked(g['labels'], g['masks'])
    pred, loss, grads = U.run_oracle(opt, params, vid, True)
    assert float(g['train|top1_gt_0.9']) > 0.6                      # the regime the case exists for
    assert abs(loss - float(g['train|loss'])) < 1e-5 * abs(float(g['train|loss']))
    s = SM.summarize_logp(pred)
    assert np.abs(s['slice'] - g['train|logp|slice']).max() < 1e-6 * np.abs(g['train|logp|slice']).max()
    assert np.array_equal(s['argmax'], g['train|logp|argmax'])
    for key, v in SM.summarize_grads(grads).items():
        ref = g['train|grad|' + key]
        assert np.allclose(v, ref, rtol=1e-4, atol=1e-7 * (1 + np.abs(ref).max())), key


def test_peaked_regime_noise_floor_of_fp32_itself():
    """What ANY fp32 evaluation of the peaked case is worth: the oracle (= the reference's arithmetic, torch CPU fp32) on 8 threads against
    itself on 1 thread -- a different summation order in the same library.  Logits reach +-200 there, so log-probs move by ~1e-4 and
    gradients by ~4e-6 of their tensor's max-norm between two CORRECT fp32 runs (against float64: 1.25e-4 / 5e-6).  The GPU gates of
    tests/test_gpu_timed_path.py for this case sit on these figures."""
    g = U.gold('case_peaked.npz')
    opt, params, vid = synth.make_peaked(g['labels'], g['masks'])
    n0 = torch.get_num_threads()
    try:
        torch.set_num_threads(max(2, min(8, n0)))
        pa, la, ga = U.run_oracle(opt, params, vid, True)
        torch.set_num_threads(1)
        pb, lb, gb = U.run_oracle(opt, params, vid, True)
    finally:
        torch.set_num_threads(n0)
    dlogp = float(np.abs(pa - pb).max())
    dgrad = max(U.relerr(ga[k], gb[k], U.GRAD_FLOOR) for k in ga if ga[k] is not None and k not in U.NOISE_ONLY)
    print('peaked case, fp32 8 threads vs 1 thread: max|dlogp| %.2e, worst gradient %.2e of its max-norm' % (dlogp, dgrad))
    assert dlogp < 1e-3 and dgrad < 3e-5 and abs(la - lb) < 1e-5 * abs(la)
    if n0 >= 2:
        assert dlogp > 1e-6          # (the case really is in the regime where fp32 summation order shows)


@pytest.mark.parametrize('case', ['tiny', 'c1', 'init', 'initc'])
def test_greedy_sample(case):
    opt, params, vid = synth.make_case(case)
    g = U.gold('case_%s.npz' % case)
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    with torch.no_grad():
        seq, lp = O.caption_forward(P, torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), None,
                                    vid['ind'], vid['soi'], 'eval', None, opt.n_head, opt.CG_seq_length, init_feats_type=opt.CG_init_feats_type)
    assert np.array_equal(seq.numpy(), g['sample|seq'])
    assert np.abs(lp.numpy() - g['sample|logp']).max() < 1e-5


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_greedy_sample_mixed_finish(name):
    """Events that emit <eos> at different steps (OldModel_NEW.py:171-183): the oracle against the reference's seq / seqLogprobs."""
    opt, params, vid = synth.make_eosmix(name)
    g = U.gold('case_eosmix.npz')
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    with torch.no_grad():
        seq, lp = O.caption_forward(P, torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), None,
                                    g[name + '|ind'], g[name + '|soi'], 'eval', None, opt.n_head, opt.CG_seq_length)
    assert np.array_equal(seq.numpy(), g[name + '|seq']) and (g[name + '|seq'] == 0).any()
    assert np.abs(lp.numpy() - g[name + '|logp']).max() < 1e-5


def test_adam_restatement():
    g = U.gold('adam.npz')
    p, m, v = g['p0'].copy(), np.zeros_like(g['p0']), np.zeros_like(g['p0'])
    for i in range(4):
        O.clamp_adam_step(p, g['g%d' % i].copy(), m, v, i + 1, 5e-5)
        assert np.abs(p - g['p%d' % (i + 1)]).max() < 1e-6


def test_proposal_selection_indices():
    g = U.gold('proposals.npz')
    for i in range(3):
        ind, feat, _ = O.top_proposals(g['g%d|scores' % i], g['g%d|mask' % i], int(g['g%d|topN' % i]))
        assert np.array_equal(np.array(ind, np.int64), g['g%d|ind' % i])
        assert np.array_equal(np.array(feat, np.int64).reshape(-1, 2), g['g%d|feat' % i].reshape(-1, 2))


def test_sst_restatement_matches_reference():
    """The oracle's hand-written 2-layer LSTM + head + weighted BCE against the reference's SST / TAPModelCriterion outputs."""
    g = U.gold('sst.npz')
    P = {k[len('param|'):]: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in g.items() if k.startswith('param|')}
    x = torch.from_numpy(g['x'])
    tap, sc = O.sst_forward(P, x)
    loss = O.tap_criterion(sc, torch.from_numpy(g['masks']), torch.from_numpy(g['labels']), torch.from_numpy(g['w1'])) + 0.1 * (tap * tap).sum()
    loss.backward()
    assert np.abs(tap.detach().numpy() - g['tap']).max() < 1e-6 and np.abs(sc.detach().numpy() - g['scores']).max() < 1e-6
    assert abs(float(loss.detach()) - float(g['loss'])) < 1e-5
    for k, p in P.items():
        assert U.relerr(p.grad.numpy(), g['grad|' + k]) < 1e-4, k


def test_oracle_nms_proposals_match_reference_fixture():
    """oracle.top_proposals_nms against the reference's gettop1000_nms outputs (tools/make_golden.py do_proposals)."""
    g = U.gold('proposals.npz')
    for i in range(3):
        pick, props, conf = O.top_proposals_nms(g['n%d|scores' % i], float(g['n%d|overlap' % i]), int(g['n%d|topN' % i]))
        assert np.array_equal(props, g['n%d|props' % i]) and np.array_equal(conf, g['n%d|conf' % i])
        assert len(pick) <= int(g['n%d|topN' % i])


def test_c5_joint_sst_and_caption_path():
    """BASELINE config 5 as one unit (SST over 256 segments -> tap_feats -> caption path on proposals of 4..256 segments, joint loss,
    gradients into both models): the oracle against the reference-generated summaries (tools/make_golden.py do_c5)."""
    from tests.util import oracle_drop
    opt, params, sst_params, vid = synth.make_c5()
    g = U.gold('case_c5.npz')
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in params.items()}
    SP = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in sst_params.items()}
    c3d, lda = torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda'])
    labels, masks = torch.from_numpy(vid['labels']), torch.from_numpy(vid['masks'])
    tap, props = O.sst_forward(SP, c3d)
    pred = O.caption_forward(P, tap, c3d, lda, labels, vid['ind'], vid['soi'], 'train', oracle_drop(opt), opt.n_head)
    tap_loss = O.tap_criterion(props, torch.from_numpy(vid['tap_masks']), torch.from_numpy(vid['tap_labels']), torch.from_numpy(vid['w1']))
    cg_loss = O.lm_criterion(pred, labels[:, 1:], masks[:, 1:])
    (opt.lambda1 * tap_loss + opt.lambda2 * cg_loss).backward()
    assert abs(float(tap_loss) - float(g['train|tap_loss'])) < 1e-4 and abs(float(cg_loss) - float(g['train|cg_loss'])) < 1e-5
    s = SM.summarize_logp(pred.detach().numpy())
    assert np.abs(s['slice'] - g['train|logp|slice']).max() < 1e-5
    for key, v in SM.summarize_grads({k: p.grad.numpy() for k, p in SP.items()}).items():
        ref = g['train|sstgrad|' + key]
        assert np.allclose(v, ref, rtol=1e-3, atol=1e-6 * (1 + np.abs(ref).max())), key


def test_topn_nms_and_reranking_match_reference_fixture():
    """oracle.topn_nms / oracle.rerank against the reference's gettopN_nms / reranking outputs (tools/make_golden.py do_proposals)."""
    g = U.gold('proposals.npz')
    for i in range(3):
        pick = O.topn_nms(g['m%d|props' % i], g['m%d|pscore' % i], g['m%d|sscore' % i], float(g['m%d|thr' % i]), int(g['m%d|topN' % i]))
        assert np.array_equal(np.array(pick, np.int64), g['m%d|pick' % i])
        assert len(pick) <= int(g['m%d|topN' % i])
    for i in range(3):
        info = [{'re_score': float(x), 'id': j} for j, x in enumerate(g['r%d|scores' % i])]
        assert [v['id'] for v in O.rerank(info)] == list(g['r%d|kept' % i])
