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
