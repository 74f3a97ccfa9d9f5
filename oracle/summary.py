"""Compact summaries of large outputs, shared by tools/make_golden.py and the parity tests.

TEST INFRASTRUCTURE (see oracle/echr_ref_cpu.py header).  Full tensors of the config-1/2 shapes
(log-probs 25.6 MB, gradients 87 MB) cannot be committed, so fixtures keep: the loss, a fixed
column slice of the log-probs, per-row argmax + top1-top2 margins, and per-parameter gradient
norms plus two small slices."""
import numpy as np


def logp_columns(V1, n=64):
    return np.unique(np.linspace(0, V1 - 1, n).astype(np.int64))


def summarize_logp(logp):
    """logp: float32 numpy [N,S,V1]."""
    N, S, V1 = logp.shape
    cols = logp_columns(V1)
    part = np.partition(logp, V1 - 2, axis=2)
    top1 = part[:, :, V1 - 1]
    top2 = part[:, :, V1 - 2]
    return dict(shape=np.array(logp.shape), cols=cols, slice=logp[:, :, cols].copy(),
                argmax=logp.argmax(2).astype(np.int64), top1=top1.copy(), margin=(top1 - top2).copy(),
                prob_sum=np.exp(logp.astype(np.float64)).sum(2))


def grad_slices(g, n=32):
    f = np.asarray(g, dtype=np.float32).reshape(-1)
    stride = max(1, f.size // n)
    return f[:n].copy(), f[::stride][:n].copy()


def summarize_grads(grads):
    """grads: dict name -> float32 numpy array (None entries are skipped)."""
    out = {}
    for name, g in grads.items():
        if g is None:
            continue
        g = np.asarray(g)
        head, strided = grad_slices(g)
        out[name + '|l2'] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out[name + '|linf'] = np.float64(np.abs(g).max())
        out[name + '|head'] = head
        out[name + '|strided'] = strided
    return out
