"""Factories with the reference's names and behaviour (models/__init__.py:6-29)."""
from .OldModel_NEW import (ShowAttendTellModel, AllImgModel, H3Model, TwostreamModel, Twostream_jump_Model,  # noqa: F401
                           ThreestreamModel, TwostreamModel_3LSTM, H3denseModel, H3denaddModel, ThreestreamModel_2stream,
                           ThreestreamModel_2stream_LDA, ThreestreamModel_2stream_CC)
from .sst_model import SST
from .MA_attention_8_NEW import MA_Attention8


def setup_lm(lm_opt):
    if lm_opt.caption_model == 'show_attend_tell':
        return ShowAttendTellModel(lm_opt)              # raises: outside the hot path
    if lm_opt.caption_model == 'three_stream':
        assert lm_opt.CG_num_layers == 3
        return ThreestreamModel(lm_opt)
    raise Exception("caption model not supported: {}".format(lm_opt.caption_model))


def setup_tap(tap_opt):
    if tap_opt.tap_model == 'SST':
        return SST(tap_opt)
    raise Exception("tap model not supported: {}".format(tap_opt.tap_model))


def setup_fusion(fusion_opt):
    if fusion_opt.fusion_model == 'TSRM8':
        return MA_Attention8(fusion_opt)
    raise Exception("fusion model not supported: {}".format(fusion_opt.fusion_model))
