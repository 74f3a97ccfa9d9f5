"""CPU: host-side mirror of the reference interface + the C-ABI library's exports (no compute calls)."""
import copy
import os
import re

import numpy as np
import pytest
import torch

import echr_amd
from echr_amd import philox, synth
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract_and_opt_side_effects():
    opt = synth.default_opt()
    m = echr_amd.CaptionGenerator(opt)
    want = synth.state_dict_shapes(opt)
    sd = m.state_dict()
    assert set(sd) == set(want)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(want[k]), k
    assert (opt.video_context_dim, opt.event_context_dim, opt.clip_context_dim) == (100, 512, 500)   # CaptionGenerator.py:82-84
    assert (opt.TSRM_input_dim, opt.d_pos_vec) == (1012, 512)                                         # MA_attention_8_NEW.py:14-22
    assert sum(v.numel() for v in sd.values()) == 11465343 + 2049 * 5001
    assert m.lm_model.seq_length == opt.CG_seq_length and m.lm_model.vocab_size == opt.CG_vocab_size and m.lm_model.ss_prob == 0.0
    assert float(m.lm_model.logit.bias.detach().abs().max()) == 0.0 and float(m.lm_model.embed.weight.detach().abs().max()) <= 0.1 + 1e-6   # uniform_(-0.1, 0.1) may round onto fp32(0.1) > 0.1


def test_loads_reference_shaped_state_dict():
    opt, params, _ = synth.make_case('tiny')
    m = echr_amd.CaptionGenerator(opt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    assert torch.equal(m.fusion_model.enc_attn.linear_out_1.weight, torch.from_numpy(params['fusion_model.enc_attn.linear_out_1.weight']))


def test_factories_and_unsupported_variants():
    from echr_amd import models
    opt = synth.default_opt()
    echr_amd.CaptionGenerator(copy.copy(opt))
    idle = models.setup_lm(synth.default_opt(caption_model='show_attend_tell', clip_context_dim=500))     # a parameter container (train_SST.sh)
    with pytest.raises(NotImplementedError):
        idle.forward()
    with pytest.raises(NotImplementedError):
        models.H3Model(opt)                                                     # the other ablation decoders stay unsupported
    with pytest.raises(Exception):
        models.setup_lm(synth.default_opt(caption_model='nope'))
    with pytest.raises(Exception):
        models.setup_fusion(synth.default_opt(fusion_model='nope'))
    sst = models.setup_tap(opt)
    assert sst.rnn.hidden_size == 512 and sst.scores.out_features == 256


def test_library_exports_every_declared_symbol():
    from echr_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'echr_hip.h')).read()
    declared = set(re.findall(r'\b(echr_[a-z0-9_]+)\s*\(', hdr))
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.echr_version() == _lib.ABI_VERSION == 3


def test_philox_known_answer_and_mask_rate():
    # Random123 known-answer vectors for philox4x32-10
    out = philox.philox4x32_10(np.array([0]), 0, 0, 0, 0, 0)
    assert [int(x[0]) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = philox.philox4x32_10(np.array([0xffffffff], dtype=np.uint64), 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    m = philox.scale_mask((64, 512), 0.5, 1, 2, philox.SITE_H1, 3)
    assert set(np.unique(m)) == {0.0, 2.0} and abs((m > 0).mean() - 0.5) < 0.02
    m3 = philox.scale_mask((64, 16, 64), 0.3, 1, 2, philox.SITE_TSRM, 0)
    assert abs((m3 > 0).mean() - 0.7) < 0.02 and abs(m3.max() - 1 / 0.7) < 1e-6


def test_static_position_helpers_match_reference_numpy():
    from echr_amd.models.MA_attention_8_NEW import MA_Attention8
    g = U.gold('position.npz')
    for k in ('a', 'b'):
        pm = MA_Attention8.extract_position_matrix(g[k + '|soi'], len(g[k + '|soi']))
        assert np.array_equal(pm, g[k + '|pos_matrix'])
        pe = MA_Attention8.extract_position_embedding(pm, 512)
        assert pe.dtype == np.float64 and np.array_equal(pe.astype(np.float32), g[k + '|pos_emb_f32'])


def test_decoder_step_count_and_clip_view():
    from echr_amd.models.OldModel_NEW import ClipView, n_decoder_steps
    from oracle import echr_ref_cpu as O
    lab = np.array([[0, 5, 6, 0, 0, 0], [0, 7, 0, 0, 0, 0]])
    assert n_decoder_steps(lab) == O.n_decoder_steps(lab) == 3
    assert n_decoder_steps(torch.from_numpy(lab)) == 3
    lab2 = np.array([[0, 5, 6, 1, 2, 0]])
    assert n_decoder_steps(lab2) == O.n_decoder_steps(lab2) == 5
    _, _, vid = synth.make_case('tiny')
    c3d = torch.from_numpy(vid['c3d'])
    soi = vid['soi']
    cv = ClipView(c3d, torch.from_numpy(soi[:, 0].astype(np.int32)), torch.from_numpy((soi[:, 1] - soi[:, 0]).astype(np.int32)),
                  int((soi[:, 1] - soi[:, 0]).max()))
    clip, mask = cv.materialize()
    rc, rm = O.clip_context(c3d, soi)
    assert torch.equal(clip, rc) and torch.equal(mask, rm)
    back = ClipView.from_padded(clip, mask)
    assert torch.equal(back.ev_len, cv.ev_len)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'echr_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+\.*oracle', src, re.M), os.path.join(dirpath, f)


def test_cpu_call_fails_loudly():
    from echr_amd._lib import EchrHipError
    opt, params, vid = synth.make_case('tiny')
    m = echr_amd.CaptionGenerator(opt)
    with pytest.raises(EchrHipError):
        m(torch.from_numpy(vid['tap']), torch.from_numpy(vid['c3d']), torch.from_numpy(vid['lda']), torch.from_numpy(vid['labels']),
          vid['ind'], vid['soi'], mode='train')


def test_reference_made_checkpoint_loads():
    """tests/golden/ref_tiny_checkpoint.pth was written by the reference's own CaptionGenerator / SST (tools/make_golden.py,
    train.py's dict layout): both state dicts load strictly into the build's modules."""
    from echr_amd import models
    opt = synth.default_opt(**synth.CASES['tiny']['opt'])
    ck = torch.load(os.path.join(U.GOLD, 'ref_tiny_checkpoint.pth'), map_location='cpu')
    assert set(ck) >= {'iteration', 'cg_model', 'tap_model'} and ck['iteration'] == 7
    cg = echr_amd.CaptionGenerator(opt)
    res = cg.load_state_dict(ck['cg_model'], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    tap = models.setup_tap(opt)
    res = tap.load_state_dict(ck['tap_model'], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(cg.lm_model.core.layer1.weight_ih, ck['cg_model']['lm_model.core.layer1.weight_ih'])


def test_host_side_proposal_utilities_match_reference_fixture():
    """echr_amd.eval_utils.gettopN_nms / reranking (host numpy like the reference's eval_utils.py:230-256, :334-345) against the
    reference-generated picks."""
    from echr_amd import eval_utils as EU
    from tests import util as U
    g = U.gold('proposals.npz')
    for i in range(3):
        props, psc, ssc = g['m%d|props' % i], g['m%d|pscore' % i], g['m%d|sscore' % i]
        rp, rs_, pick = EU.gettopN_nms(props, psc, ssc, nms_overlap=float(g['m%d|thr' % i]), topN=int(g['m%d|topN' % i]))
        assert np.array_equal(np.asarray(pick, np.int64), g['m%d|pick' % i])
        assert np.array_equal(rp, props[g['m%d|pick' % i]]) and np.array_equal(rs_, psc[g['m%d|pick' % i]])
    for i in range(3):
        info = [{'re_score': float(x), 'id': j} for j, x in enumerate(g['r%d|scores' % i])]
        assert [v['id'] for v in EU.reranking(info)] == list(g['r%d|kept' % i])


def test_show_attend_tell_recipe_builds_a_state_dict_compatible_container():
    """experiments/train_SST.sh:4 builds cg_model with caption_model='show_attend_tell' (CG_num_layers 3, ER3 / VL / CC) while only the proposal
    encoder trains (train.py:291-295).  The drop-in constructs it, with the reference's state_dict names and shapes (printed from the
    reference: models/OldModel_NEW.py:190-216,1009-1012), and refuses to run it."""
    import echr_amd
    opt = synth.default_opt(vocab_size=30, seq_length=5, caption_model='show_attend_tell', CG_num_layers=3)
    m = echr_amd.CaptionGenerator(opt)
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    expect = {'lm_model.embed.weight': (31, 512), 'lm_model.logit.weight': (31, 512), 'lm_model.logit.bias': (31,),
              'lm_model.core.rnn.weight_ih_l0': (2048, 512), 'lm_model.core.rnn.weight_hh_l0': (2048, 512),
              'lm_model.core.rnn.weight_ih_l1': (2048, 512), 'lm_model.core.rnn.weight_hh_l1': (2048, 512),
              'lm_model.core.rnn.weight_ih_l2': (2048, 512), 'lm_model.core.rnn.weight_hh_l2': (2048, 512),
              'lm_model.core.ctx2att.weight': (512, 500), 'lm_model.core.ctx2att.bias': (512,),
              'lm_model.core.h2att.weight': (512, 512), 'lm_model.core.h2att.bias': (512,),
              'lm_model.core.alpha_net.weight': (1, 512), 'lm_model.core.alpha_net.bias': (1,)}
    lm = {k: v for k, v in sd.items() if k.startswith('lm_model.')}
    assert lm == expect, set(lm.items()) ^ set(expect.items())
    assert any(k.startswith('fusion_model.') for k in sd)                       # ER3 + TSRM8: the event encoder is built as in the reference
    m.load_state_dict(m.state_dict())
    with pytest.raises(NotImplementedError):
        m(torch.zeros(4, 512), torch.zeros(4, 500), torch.zeros(100), torch.zeros(1, 3, dtype=torch.long), [1], [[0, 2]], mode='train')
    opt2 = synth.default_opt(vocab_size=30, seq_length=5, caption_model='show_attend_tell', CG_num_layers=2, CG_input_feats_type='E')
    assert tuple(echr_amd.CaptionGenerator(opt2).state_dict()['lm_model.core.rnn.weight_ih_l0'].shape) == (2048, 1024)


def test_three_stream_core_accepts_and_ignores_input_feats_type():
    """The reference's ThreeStream_Core computes CG_input_dim from CG_input_feats_type and never uses it (models/OldModel_NEW.py:775-776,
    :790-799; forward :801-823 reads neither): same state_dict, same attribute values here -- no raise."""
    import echr_amd
    base = echr_amd.CaptionGenerator(synth.default_opt(vocab_size=30, seq_length=5))
    opt = synth.default_opt(vocab_size=30, seq_length=5, CG_input_feats_type='VEC')
    m = echr_amd.CaptionGenerator(opt)
    core = m.lm_model.core
    assert core.CG_input_feats_type == 'VEC'
    assert core.CG_input_dim == opt.video_context_dim + opt.event_context_dim + opt.clip_context_dim
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in base.state_dict().items()}


def test_persistent_role_placement_is_a_bijection():
    """csrc/persist.hip persist_role_index: under every placement each of the 256 workgroups of a merged recurrence launch plays exactly one
    role (0..95 half machine 0, 96..191 half machine 1, 192..255 the two LSTM streams), a half machine sits on three XCDs (dispatch position
    b % 8) and an LSTM stream on one; mode 2 keeps the 64 product roles (16..79) of a half machine on two of its XCDs."""
    from echr_amd import _lib
    lib = _lib.load()
    for mode in (0, 1, 2):
        roles = [lib.echr_persist_role_index(b, mode) for b in range(256)]
        assert sorted(roles) == list(range(256)), mode
    for mode in (1, 2):
        xcds = {}
        for b in range(256):
            r = lib.echr_persist_role_index(b, mode)
            group = r // 96 if r < 192 else 2 + (r - 192) // 32
            xcds.setdefault(group, set()).add(b % 8)
        assert [len(xcds[g]) for g in range(4)] == [3, 3, 1, 1], (mode, xcds)
        assert not (xcds[0] & xcds[1]) and not (xcds[2] & xcds[3]) and not ((xcds[0] | xcds[1]) & (xcds[2] | xcds[3]))
    prod = {}
    for b in range(256):
        r = lib.echr_persist_role_index(b, 2)
        if r < 192 and 16 <= r % 96 < 80:
            prod.setdefault(r // 96, set()).add(b % 8)
    assert all(len(v) == 2 for v in prod.values()), prod


def test_runtime_switches_are_known_by_name():
    """echr_config_set (include/echr_hip.h): the switches the documentation names are accepted (a host-side table, no device call), the
    fixed-order accumulation switch among them (`echr_amd.set_deterministic`), and an unknown key is an error with a message -- never a silent
    no-op."""
    import echr_amd
    from echr_amd import _lib
    lib = _lib.load()
    for key, val in ((b'deterministic', 1), (b'deterministic', 0), (b'persist', 1), (b'persist_bwd', 1), (b'gemm_h2', 1), (b'gemm_bf16x3', 1), (b'att_slots', 2)):
        assert lib.echr_config_set(key, val) == 0, key
    echr_amd.set_deterministic(True)
    echr_amd.set_deterministic(False)
    assert lib.echr_config_set(b'no_such_switch', 1) == -22
    assert b'no_such_switch' in lib.echr_last_error()
    assert lib.echr_config_set(b'att_slots', 3) != 0          # (only 2, 4 or 8)
