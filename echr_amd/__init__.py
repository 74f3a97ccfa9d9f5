"""echr_amd -- MI355X (gfx950) native implementation of ECHR's hierarchical encoder + attention caption
decoder hot path, behind the reference's own module API (CaptionGenerator, models.setup_lm / setup_fusion /
MA_attention_8_NEW).  All arithmetic lives in libechr_hip.so (include/echr_hip.h); there is no CPU fallback."""
from .CaptionGenerator import CaptionGenerator  # noqa: F401
from . import models  # noqa: F401


def set_deterministic(on=True):
    """Fixed-order accumulation for every sum of a training iteration (`echr_config_set("deterministic", ...)`, include/echr_hip.h): two runs on the
    same inputs, parameters and dropout seed agree bit for bit in loss and gradients, as the reference's CPU path does at a fixed thread count.
    Slower than the default (atomic split-K, persistent recurrences); returns nothing, raises if the library is missing."""
    from . import _lib
    _lib.check(_lib.load().echr_config_set(b'deterministic', 1 if on else 0), 'config_set')
