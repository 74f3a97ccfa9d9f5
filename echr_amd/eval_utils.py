"""Proposal selection for evaluation -- the index-producing part of the reference's eval_utils.py.

`gettop1000` keeps the reference's signature and return tuple (eval_utils.py:259-287) but runs the threshold search and the
ordered enumeration in one HIP kernel (echr_top_proposals) instead of a numpy sort + an O(T*K) python double loop; the
device tensors it produced are also returned by `top_proposals_device` so the caption path can consume them without a
round trip through python lists.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .misc import utils


def top_proposals_device(pred_proposals, tap_masks, topN=1000, val_score_thres=0.0):
    """(ind [M] int32, featstamps [M,2] int32, confidence [M] fp32) device tensors; one D2H sync for M."""
    lib = L.load()
    dev = pred_proposals.device
    scores = pred_proposals.detach().to(torch.float32).contiguous()
    masks = torch.as_tensor(np.asarray(tap_masks) if not isinstance(tap_masks, torch.Tensor) else tap_masks).to(dev, torch.float32).contiguous()
    T, K = scores.shape
    ind = torch.empty(T * K, device=dev, dtype=torch.int32)
    feat = torch.empty(T * K, 2, device=dev, dtype=torch.int32)
    conf = torch.empty(T * K, device=dev, dtype=torch.float32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    L.check(lib.echr_top_proposals(L.ptr(scores), L.ptr(masks), T, K, int(topN), float(val_score_thres), L.ptr(ind, torch.int32),
                                   L.ptr(feat, torch.int32), L.ptr(conf), L.ptr(cnt, torch.int32), L.stream_ptr()), 'top_proposals')
    m = int(cnt.item())
    return ind[:m], feat[:m], conf[:m]


def gettop1000(pred_proposals, tap_masks, cg_gts, duration, featstamp_to_time, val_score_thres=0, topN=1000):
    """Same outputs as the reference: (index_select_list, featstamp_list, cg_select_list, timestamp_list, confidence)."""
    if not isinstance(pred_proposals, torch.Tensor):
        pred_proposals = torch.as_tensor(np.asarray(pred_proposals, dtype=np.float32))
    if not pred_proposals.is_cuda:
        raise L.EchrHipError('gettop1000 runs on the GPU: pass the SST scores as a CUDA tensor')
    nfeats = pred_proposals.shape[0]
    ind, feat, conf = top_proposals_device(pred_proposals, tap_masks, topN, float(val_score_thres))
    ind_l, feat_l, conf_l = ind.cpu().tolist(), feat.cpu().tolist(), conf.cpu().tolist()
    cg_l = [cg_gts[n, n - s] for n, (s, _) in zip(ind_l, feat_l)] if len(cg_gts) else []
    time_l = [featstamp_to_time(s, e, nfeats, duration) for s, e in feat_l]
    return ind_l, feat_l, cg_l, time_l, conf_l


def gettop1000_nms(pred_proposals, tap_masks, cg_gts, duration, featstamp_to_time, overlap=0.8, topN=1000):
    """Same outputs as the reference (eval_utils.py:290-331): (index_select_list, nms_props [M,2], prop_gts, timestamp_list,
    nms_scores), with the candidate enumeration, the score ordering and the greedy suppression in one HIP kernel
    (echr_top_proposals_nms).  `tap_masks` is unused, as in the reference."""
    lib = L.load()
    if not isinstance(pred_proposals, torch.Tensor):
        pred_proposals = torch.as_tensor(np.asarray(pred_proposals, dtype=np.float32))
    if not pred_proposals.is_cuda:
        raise L.EchrHipError('gettop1000_nms runs on the GPU: pass the SST scores as a CUDA tensor')
    scores = pred_proposals.detach().to(torch.float32).contiguous()
    dev = scores.device
    T, K = scores.shape
    scratch = torch.empty(T * K, device=dev, dtype=torch.float32)
    feat = torch.empty(int(topN), 2, device=dev, dtype=torch.int32)
    conf = torch.empty(int(topN), device=dev, dtype=torch.float32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    L.check(lib.echr_top_proposals_nms(L.ptr(scores), T, K, int(topN), float(overlap), L.ptr(scratch), L.ptr(feat, torch.int32), L.ptr(conf),
                                       L.ptr(cnt, torch.int32), L.stream_ptr()), 'top_proposals_nms')
    m = int(cnt.item())
    props = feat[:m].cpu().numpy().astype(np.int64)
    nms_scores = conf[:m].cpu().numpy().astype(np.float64)
    prop_gts = np.array([cg_gts[e - 1, e - 1 - s] for s, e in props]) if len(cg_gts) else np.array([])
    timestamp_list = [featstamp_to_time(s, e, T, duration) for (s, e) in props]
    return props[:, 1] - 1, props, prop_gts, timestamp_list, nms_scores


def caption_video(tap_model, cg_model, c3d_feats, lda_feats, duration, featstamp_to_time, vocab=None, tap_masks=None, cg_gts=(),
                  topN=1000, nms_threshold=0.0, val_score_thres=0.0, flag_eval_what='tap_cg'):
    """One video through the reference's evaluation flow (eval_utils.py:51-53,106-167 for flag_eval_what 'tap_cg' / 'tap'):
    SST -> proposal selection (greedy NMS when nms_threshold != 0, else score threshold) -> greedy captions -> the per-proposal
    records of result.json.  Everything between the two host reads (proposal count, caption lengths) stays on the GPU.

    Returns (vid_info, extras): vid_info is the reference's list of dicts (sentence, timestamp, sentence_confidence, proposal_score,
    re_score, num); extras carries the tensors (tap_feats, pred_proposals, seq, ind_select_list, soi_select_list)."""
    if not c3d_feats.is_cuda:
        raise L.EchrHipError('caption_video runs on the GPU: move the models and features with .cuda()')
    nfeats = c3d_feats.shape[0]
    with torch.no_grad():
        tap_feats, pred_proposals = tap_model(c3d_feats)
        if tap_masks is None:
            K = pred_proposals.shape[1]
            tap_masks = (np.arange(nfeats)[:, None] >= np.arange(K)[None, :]).astype(np.float32)
        if nms_threshold != 0:
            ind_select_list, soi_select_list, cg_select_list, good_time_stamps, tap_prob = gettop1000_nms(
                pred_proposals, tap_masks, cg_gts, duration, featstamp_to_time, overlap=nms_threshold, topN=topN)
        else:
            ind_select_list, soi_select_list, cg_select_list, good_time_stamps, tap_prob = gettop1000(
                pred_proposals, tap_masks, cg_gts, duration, featstamp_to_time, val_score_thres=val_score_thres, topN=topN)
        extras = dict(tap_feats=tap_feats, pred_proposals=pred_proposals, ind_select_list=ind_select_list, soi_select_list=soi_select_list,
                      seq=None, cg_prob=None)
        n = len(ind_select_list)
        if n == 0:
            return [], extras
        if flag_eval_what == 'tap':
            sents, cg_score = [0] * n, [0] * n
        else:
            seq, cg_prob = cg_model(tap_feats, c3d_feats, lda_feats, [], ind_select_list, soi_select_list, mode='eval')
            if len(seq) == 0:
                return [], extras
            extras['seq'], extras['cg_prob'] = seq, cg_prob
            cg_score = cg_prob.sum(1).cpu().numpy().astype('float')
            sents = utils.decode_sequence(vocab, seq) if vocab is not None else [row[row > 0].tolist() for row in seq.cpu().numpy()]
    vid_info = []
    for i, sent in enumerate(sents):
        vid_info.append({'sentence': sent, 'timestamp': good_time_stamps[i], 'sentence_confidence': cg_score[i],
                         'proposal_score': float(tap_prob[i]), 're_score': 10 * float(tap_prob[i]) + cg_score[i], 'num': [i, len(sents)]})
    return vid_info, extras


def gettopN_nms(props, prop_scores, sent_score, nms_overlap=0.999, topN=1000):
    """Host-side greedy temporal NMS over already materialised proposals (eval_utils.py:230-256; called from the per-video loop at
    :89 with sent_score = prop_scores).  Proposals are visited by descending `prop_scores`; every proposal whose IoU (with the
    reference's +1e-3 interval closure) with the current best reaches `nms_overlap` forms its cluster, the cluster is represented by
    its member with the highest `sent_score`, and only proposals with IoU <= nms_overlap stay for the next round.  Returns
    (props[pick], prop_scores[pick], pick) like the reference.  A handful of proposals per video: numpy on the host, as there."""
    props = np.asarray(props)
    prop_scores = np.asarray(prop_scores)
    sent_score = np.asarray(sent_score)
    start, end = props[:, 0], props[:, 1]
    length = (end - start + 1e-3).astype(float)
    alive = np.argsort(prop_scores)                 # ascending; the current best is the last entry (same sort call as the reference)
    pick = []
    while alive.size and len(pick) < topN:
        best = alive[-1]
        inter = np.maximum(0., np.minimum(end[best], end[alive]) - np.maximum(start[best], start[alive]) + 1e-3)
        iou = inter / (length[best] + length[alive] - inter)
        cluster = alive[np.nonzero(iou >= nms_overlap)[0]]
        pick.append(cluster[np.argmax(sent_score[cluster])])
        alive = alive[np.nonzero(iou <= nms_overlap)[0]]
    return props[pick, :], prop_scores[pick], pick


def reranking(vid_info):
    """Keep the videos whose 're_score' reaches the 10th largest one (all of them when there are fewer than 10): eval_utils.py:334-345."""
    scores = np.sort(np.array([v['re_score'] for v in vid_info]))
    threshold = scores[-min(len(scores), 10)]
    return [v for v in vid_info if v['re_score'] >= threshold]
