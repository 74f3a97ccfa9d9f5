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
