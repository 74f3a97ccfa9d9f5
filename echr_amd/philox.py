"""Counter-based dropout masks (Philox-4x32-10), host side.

The decoder's dropout sites (reference: models/OldModel_NEW.py:810,814,818 for the three
stream outputs, :136 for the late-fusion input, models/MA_attention_8_NEW.py:162 for the
event-relation attention) draw their keep/drop decisions from one counter-based generator so
that (i) the backward kernels regenerate the forward's mask instead of storing it and (ii) the
CPU oracle can be fed bit-identical masks in parity tests.  The device implementation lives in
csrc/echr_common.h (`echr_keep`); this numpy version must stay bit-identical to it.

Counter layout (4 x u32):  c0 = element_index >> 2, c1 = step, c2 = site, c3 = offset
Key (2 x u32):             k0 = seed & 0xffffffff, k1 = seed >> 32
Element e uses output word (e & 3).  keep  <=>  word >= floor(p * 2**32).
"""
import numpy as np

SITE_TSRM = 0      # [N, G, N]  p = 0.3
SITE_H0 = 1        # [N, H]     p = 0.5 (stream 0 output)
SITE_H1 = 2
SITE_H2 = 3
SITE_OUT = 4       # [N, 3H]    p = CG_drop_prob
SITE_SST = 5       # [T, H]     p = rnn_dropout (SST inter-layer dropout; step counter 0, element t*H + j)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  Inputs: u32 arrays (broadcastable); returns 4 u32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64) + np.zeros_like(c0)
    c2 = np.asarray(c2, dtype=np.uint64) + np.zeros_like(c0)
    c3 = np.asarray(c3, dtype=np.uint64) + np.zeros_like(c0)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = (hi1 ^ c1 ^ np.uint64(k0)) & _MASK
        n1 = lo1
        n2 = (hi0 ^ c3 ^ np.uint64(k1)) & _MASK
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def drop_threshold(p):
    """u32 threshold: an element is kept iff its random word >= threshold."""
    return int(np.floor(float(p) * 4294967296.0)) & 0xFFFFFFFF if p < 1.0 else 0xFFFFFFFF


def keep_mask(numel, p, seed, offset, site, step):
    """Boolean keep mask for `numel` elements (flat, row-major element index)."""
    e = np.arange(numel, dtype=np.uint64)
    words = philox4x32_10(e >> np.uint64(2), step, site, offset, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(words, axis=-1)                       # [numel, 4]
    sel = w[np.arange(numel), (e & np.uint64(3)).astype(np.int64)]
    return sel >= np.uint32(drop_threshold(p))


def scale_mask(shape, p, seed, offset, site, step):
    """float32 multiplicative mask (0 or 1/(1-p)) of the given shape."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return np.ones(shape, dtype=np.float32)
    k = keep_mask(n, p, seed, offset, site, step)
    s = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return (k.astype(np.float32) * s).reshape(shape)
