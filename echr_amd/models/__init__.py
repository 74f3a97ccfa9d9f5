"""Model factories of the ECHR path: `setup_lm`, `setup_tap`, `setup_fusion` dispatch on the option strings the reference's
drivers pass (reference: models/__init__.py:6-29).  Only the ECHR recipe is backed by HIP kernels; every other option value
raises."""
from . import OldModel_NEW as _decoder
from .MA_attention_8_NEW import MA_Attention8
from .sst_model import SST

# decoder classes the reference's package namespace exposes; only ThreestreamModel is live on the HIP path
_DECODER_NAMES = ('ShowAttendTellModel', 'AllImgModel', 'H3Model', 'TwostreamModel', 'Twostream_jump_Model', 'ThreestreamModel',
                  'TwostreamModel_3LSTM', 'H3denseModel', 'H3denaddModel', 'ThreestreamModel_2stream',
                  'ThreestreamModel_2stream_LDA', 'ThreestreamModel_2stream_CC')
globals().update({n: getattr(_decoder, n) for n in _DECODER_NAMES})

_CAPTION_MODELS = {'three_stream': 'ThreestreamModel', 'show_attend_tell': 'ShowAttendTellModel'}
_PROPOSAL_MODELS = {'SST': SST}
_FUSION_MODELS = {'TSRM8': MA_Attention8}


def _pick(table, key, what):
    if key not in table:
        raise Exception('%s not supported: %s' % (what, key))
    return table[key]


def setup_lm(lm_opt):
    """Caption decoder for `opt.caption_model` ('three_stream' needs CG_num_layers == 3, as in the reference)."""
    cls = getattr(_decoder, _pick(_CAPTION_MODELS, lm_opt.caption_model, 'caption model'))
    if lm_opt.caption_model == 'three_stream' and lm_opt.CG_num_layers != 3:
        raise AssertionError('three_stream expects CG_num_layers == 3')
    return cls(lm_opt)


def setup_tap(tap_opt):
    """Proposal encoder for `opt.tap_model`."""
    return _pick(_PROPOSAL_MODELS, tap_opt.tap_model, 'tap model')(tap_opt)


def setup_fusion(fusion_opt):
    """Event-relation encoder for `opt.fusion_model`."""
    return _pick(_FUSION_MODELS, fusion_opt.fusion_model, 'fusion model')(fusion_opt)
