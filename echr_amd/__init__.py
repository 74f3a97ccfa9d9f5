"""echr_amd -- MI355X (gfx950) native implementation of ECHR's hierarchical encoder + attention caption
decoder hot path, behind the reference's own module API (CaptionGenerator, models.setup_lm / setup_fusion /
MA_attention_8_NEW).  All arithmetic lives in libechr_hip.so (include/echr_hip.h); there is no CPU fallback."""
from .CaptionGenerator import CaptionGenerator  # noqa: F401
from . import models  # noqa: F401
