"""ctypes binding of libechr_hip.so (C ABI declared in include/echr_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ECHR_LIB: a differently built library for same-box A/B runs of kernel variants -- tools/ab_lib.sh; never set in production)
LIB_PATH = os.environ.get('ECHR_LIB') or os.path.join(_HERE, 'lib', 'libechr_hip.so')

c_f = C.c_void_p   # device pointers travel as void*
i32, i64, f32 = C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [('A', c_f), ('B', c_f), ('C', c_f), ('M', i32), ('N', i32), ('K', i32),
                ('sam', i64), ('sak', i64), ('sbk', i64), ('sbn', i64), ('ldc', i64),
                ('batch', i32), ('bsa', i64), ('bsb', i64), ('bsc', i64), ('alpha', f32), ('beta', f32),
                ('bias', c_f), ('bs_bias', i64), ('bias2', c_f), ('addend', c_f), ('add_mod', i32), ('ld_add', i64),
                ('act', i32), ('aux', c_f), ('ld_aux', i64), ('rowmap_mod', i32), ('rowmap_mul', i32), ('split_k', i32), ('algo', i32),
                ('row_index', c_f), ('row_index_max', i32)]


class Dropout(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('offset', C.c_uint32), ('training', i32),
                ('p_tsrm', f32), ('p_h', f32), ('p_out', f32)]


class TsrmArgs(C.Structure):
    _fields_ = [('N', i32), ('Din', i32), ('Df', i32), ('Do', i32), ('G', i32),
                ('w_emb', c_f), ('b_emb', c_f), ('w_fc1', c_f), ('b_fc1', c_f), ('w_fc2', c_f), ('b_fc2', c_f),
                ('w_q', c_f), ('b_q', c_f), ('w_k', c_f), ('b_k', c_f), ('w_out', c_f), ('b_out', c_f),
                ('ech', c_f), ('ev_start', c_f), ('ev_len', c_f), ('ws', c_f), ('out', c_f), ('inference', i32), ('max_len', i32), ('max_span', i32),
                ('fst_mode', i32)]


class TsrmGrads(C.Structure):
    _fields_ = [('g_w_emb', c_f), ('g_b_emb', c_f), ('g_w_fc1', c_f), ('g_b_fc1', c_f), ('g_w_fc2', c_f), ('g_b_fc2', c_f),
                ('g_w_q', c_f), ('g_b_q', c_f), ('g_w_k', c_f), ('g_b_k', c_f), ('g_w_out', c_f), ('g_b_out', c_f),
                ('g_ech', c_f), ('g_out', c_f), ('ws_bwd', c_f), ('zeroed', i32)]


class DecArgs(C.Structure):
    _fields_ = [('N', i32), ('A', i32), ('Tv', i32), ('D', i32), ('H', i32), ('E', i32), ('Ha', i32), ('De', i32),
                ('Dv', i32), ('V1', i32), ('S', i32), ('rows_disjoint', i32),
                ('embed', c_f), ('w_logit', c_f), ('b_logit', c_f),
                ('w_ih', c_f * 3), ('w_hh', c_f * 3), ('b_ih', c_f * 3), ('b_hh', c_f * 3),
                ('w_c2a', c_f), ('b_c2a', c_f), ('w_h2a', c_f), ('b_h2a', c_f), ('w_alpha', c_f), ('b_alpha', c_f),
                ('c3d', c_f), ('ev_start', c_f), ('ev_len', c_f), ('event', c_f), ('video', c_f), ('tokens', c_f),
                ('ws', c_f), ('logp', c_f), ('prepared', i32), ('train', i32), ('zero_extra', c_f), ('zero_extra_count', i64), ('h0', c_f)]


class DecGrads(C.Structure):
    _fields_ = [('g_embed', c_f), ('g_w_logit', c_f), ('g_b_logit', c_f),
                ('g_w_ih', c_f * 3), ('g_w_hh', c_f * 3), ('g_b_ih', c_f * 3), ('g_b_hh', c_f * 3),
                ('g_w_c2a', c_f), ('g_b_c2a', c_f), ('g_w_h2a', c_f), ('g_b_h2a', c_f), ('g_w_alpha', c_f), ('g_b_alpha', c_f),
                ('g_event', c_f), ('g_video', c_f), ('g_logp', c_f),
                ('nll_target', c_f), ('nll_mask', c_f), ('g_loss', c_f), ('ws_bwd', c_f), ('zeroed', i32), ('phase', i32), ('async_tail', i32), ('nll_msum', c_f),
                ('zero_extra', c_f), ('zero_extra_count', i64), ('nll_target_i64', i32), ('dlg_ready', i32), ('active_rows', c_f), ('n_active', i32), ('g_h0', c_f)]


class InitStateArgs(C.Structure):
    _fields_ = [('N', i32), ('Dv', i32), ('De', i32), ('D', i32), ('H3', i32), ('use_v', i32), ('use_e', i32), ('use_c', i32), ('A', i32),
                ('video', c_f), ('event', c_f), ('c3d', c_f), ('ev_start', c_f), ('ev_len', c_f), ('w', c_f), ('b', c_f), ('feats', c_f), ('h0', c_f)]


class InitStateGrads(C.Structure):
    _fields_ = [('g_h0', c_f), ('g_w', c_f), ('g_b', c_f), ('g_video', c_f), ('g_event', c_f), ('dfeats', c_f), ('zeroed', i32)]


class SstArgs(C.Structure):
    _fields_ = [('T', i32), ('D', i32), ('H', i32), ('K', i32), ('p_drop', f32),
                ('w_ih', c_f * 2), ('w_hh', c_f * 2), ('b_ih', c_f * 2), ('b_hh', c_f * 2), ('w_sc', c_f), ('b_sc', c_f),
                ('x', c_f), ('ws', c_f), ('tap_feats', c_f), ('scores', c_f)]


class SstGrads(C.Structure):
    _fields_ = [('g_w_ih', c_f * 2), ('g_w_hh', c_f * 2), ('g_b_ih', c_f * 2), ('g_b_hh', c_f * 2), ('g_w_sc', c_f), ('g_b_sc', c_f),
                ('g_tap', c_f), ('g_scores', c_f), ('ws_bwd', c_f), ('zeroed', i32)]


class SampleArgs(C.Structure):
    _fields_ = [('dec', DecArgs), ('seq_len', i32), ('seq', c_f), ('seq_logp', c_f), ('n_unfinished', c_f),
                ('ws_sample', c_f), ('multinomial', i32), ('temperature', C.c_float), ('seed', C.c_uint64),
                ('tables', c_f), ('tables_valid', i32)]


HANDOVER_FN = C.CFUNCTYPE(None, i32, C.c_void_p, C.c_void_p)          # echr_handover_fn
MID_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)                    # echr_mid_fn


class TrainStepArgs(C.Structure):
    _fields_ = [('tsrm', TsrmArgs), ('tsrm_g', TsrmGrads), ('dec', DecArgs), ('dec_g', DecGrads), ('drop', Dropout),
                ('tap', c_f), ('Ht', i32), ('g_tap', c_f), ('host_index', C.c_void_p),
                ('nll_target', c_f), ('nll_target_i64', i32), ('nll_mask', c_f), ('g_loss', c_f), ('loss', c_f),
                ('ws', c_f), ('ws_floats', i64), ('flat_g', c_f), ('n_flat', i64),
                ('flat_p', c_f), ('adam_m', c_f), ('adam_v', c_f), ('adam_step', i32),
                ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double), ('clip', f32),
                ('do_step', i32), ('overlap_encoder', i32), ('forward_only', i32), ('n_active', i32), ('host_nll', i32), ('prepared', i32), ('defer_update', i32),
                ('handover', i32), ('handover_cb', C.c_void_p), ('handover_user', C.c_void_p), ('adam_applied', c_f),
                ('event_parts', i32), ('w_init', c_f), ('b_init', c_f), ('g_w_init', c_f), ('g_b_init', c_f),
                ('init_use_v', i32), ('init_use_e', i32), ('init_use_c', i32), ('vh_offset', i32), ('tap_rows', i32),
                ('mid_cb', C.c_void_p), ('mid_user', C.c_void_p)]


# every symbol include/echr_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('echr_version', i32, []),
    ('echr_abi_sizeof', i64, [C.c_char_p]),
    ('echr_last_error', C.c_char_p, []),
    ('echr_check_async', i32, []),
    ('echr_async_skipped_updates', i64, []),
    ('echr_gemm_f32', i32, [C.POINTER(GemmDesc), C.c_void_p]),
    ('echr_event_pool_gather_fwd', i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_event_pool_gather_bwd', i32, [c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_init_state_fwd', i32, [C.POINTER(InitStateArgs), C.c_void_p]),
    ('echr_init_state_bwd', i32, [C.POINTER(InitStateArgs), C.POINTER(InitStateGrads), C.c_void_p]),
    ('echr_col_mean_fwd', i32, [c_f, i32, i32, i64, c_f, C.c_void_p]),
    ('echr_col_mean_bwd', i32, [c_f, i32, i32, i64, c_f, C.c_void_p]),
    ('echr_tsrm_ws_floats', i64, [i32, i32, i32, i32, i32]),
    ('echr_tsrm_ws_bwd_floats', i64, [i32, i32, i32, i32, i32]),
    ('echr_tsrm_fwd', i32, [C.POINTER(TsrmArgs), C.POINTER(Dropout), C.c_void_p]),
    ('echr_tsrm_bwd', i32, [C.POINTER(TsrmArgs), C.POINTER(TsrmGrads), C.POINTER(Dropout), C.c_void_p]),
    ('echr_tsrm_posemb', i32, [c_f, c_f, c_f, i32, i32, C.c_void_p]),
    ('echr_decoder_ws_floats', i64, [C.POINTER(DecArgs)]),
    ('echr_decoder_ws_bwd_floats', i64, [C.POINTER(DecArgs)]),
    ('echr_decoder_fwd', i32, [C.POINTER(DecArgs), C.POINTER(Dropout), C.c_void_p]),
    ('echr_decoder_fwd_prepare', i32, [C.POINTER(DecArgs), C.c_void_p]),
    ('echr_decoder_fwd_prepare_cancel', i32, [C.c_void_p]),
    ('echr_decoder_bwd', i32, [C.POINTER(DecArgs), C.POINTER(DecGrads), C.POINTER(Dropout), C.c_void_p]),
    ('echr_nll_loss_fwd', i32, [c_f, c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_nll_loss_bwd', i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_nll_loss_fwd_i64', i32, [c_f, c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_nll_loss_bwd_i64', i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, C.c_void_p]),
    ('echr_sampler_ws_floats', i64, [C.POINTER(DecArgs)]),
    ('echr_sampler_table_floats', i64, [C.POINTER(DecArgs)]),
    ('echr_decoder_sample', i32, [C.POINTER(SampleArgs), C.c_void_p]),
    ('echr_config_set', i32, [C.c_char_p, i32]),
    ('echr_stream_join', i32, [C.c_void_p]),
    ('echr_streams_init', i32, []),
    ('echr_decoder_step', i32, [C.POINTER(DecArgs), c_f, c_f, c_f, c_f, C.POINTER(Dropout), C.c_void_p]),
    ('echr_tsrm_attn_fwd', i32, [C.POINTER(TsrmArgs), c_f, c_f, C.POINTER(Dropout), C.c_void_p]),
    ('echr_persist_read_stamps', i32, [C.c_void_p, i32]),
    ('echr_persist_role_index', i32, [i32, i32]),
    ('echr_prof_enable', i32, [i32]),
    ('echr_prof_event_overhead', i32, [C.POINTER(C.c_double), C.POINTER(i64)]),
    ('echr_prof_read', i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]),
    ('echr_sst_ws_floats', i64, [i32, i32, i32, i32]),
    ('echr_sst_ws_bwd_floats', i64, [i32, i32, i32, i32]),
    ('echr_sst_fwd', i32, [C.POINTER(SstArgs), C.POINTER(Dropout), C.c_void_p]),
    ('echr_sst_fwd_states', i32, [C.POINTER(SstArgs), C.POINTER(Dropout), C.c_void_p]),
    ('echr_sst_head_fwd', i32, [C.POINTER(SstArgs), C.c_void_p]),
    ('echr_sst_bwd', i32, [C.POINTER(SstArgs), C.POINTER(SstGrads), C.POINTER(Dropout), C.c_void_p]),
    ('echr_tap_bce_fwd', i32, [c_f, c_f, c_f, c_f, c_f, i32, i32, C.c_void_p]),
    ('echr_tap_bce_fwd_ws', i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, C.c_void_p]),
    ('echr_tap_bce_bwd', i32, [c_f, c_f, c_f, c_f, c_f, c_f, i32, i32, C.c_void_p]),
    ('echr_h2_bytes', i64, [i32, i32]),
    ('echr_h2_pack', i32, [c_f, i32, i32, i64, i64, C.c_void_p, C.c_void_p]),
    ('echr_top_proposals', i32, [c_f, c_f, i32, i32, i32, f32, c_f, c_f, c_f, c_f, C.c_void_p]),
    ('echr_top_proposals_nms', i32, [c_f, i32, i32, i32, C.c_double, c_f, c_f, c_f, c_f, C.c_void_p]),
    ('echr_train_step_ws_floats', i64, [C.POINTER(TrainStepArgs)]),
    ('echr_train_step', i32, [C.POINTER(TrainStepArgs), C.c_void_p]),
    ('echr_train_step_prepare', i32, [C.POINTER(TrainStepArgs), C.c_void_p]),
    ('echr_handover_wait', i32, [i32, C.c_void_p]),
    ('echr_clamp', i32, [c_f, i64, f32, C.c_void_p]),
    ('echr_clamp_adam', i32, [c_f, c_f, c_f, c_f, i64, i32, C.c_double, C.c_double, C.c_double, C.c_double, f32, C.c_void_p]),
    ('echr_clamp_adam_counted', i32, [c_f, c_f, c_f, c_f, i64, i32, C.c_double, C.c_double, C.c_double, C.c_double, f32, c_f, C.c_void_p]),
]

ABI_STRUCTS = {'echr_gemm_desc': GemmDesc, 'echr_dropout': Dropout, 'echr_tsrm_args': TsrmArgs, 'echr_tsrm_grads': TsrmGrads,
               'echr_dec_args': DecArgs, 'echr_dec_grads': DecGrads, 'echr_sample_args': SampleArgs, 'echr_sst_args': SstArgs,
               'echr_sst_grads': SstGrads, 'echr_train_step_args': TrainStepArgs, 'echr_init_state_args': InitStateArgs, 'echr_init_state_grads': InitStateGrads}

ABI_VERSION = 3          # include/echr_hip.h ECHR_ABI_VERSION
_lib = None


class EchrHipError(RuntimeError):
    pass


def load():
    """Load libechr_hip.so (built in-tree by `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EchrHipError('libechr_hip.so not found at %s -- build it first (python __graft_entry__.py build); '
                           'echr_amd has no CPU or PyTorch fallback path' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.echr_version() != ABI_VERSION:
        raise EchrHipError('libechr_hip.so ABI version %d != %d -- rebuild the library (python __graft_entry__.py build)' % (lib.echr_version(), ABI_VERSION))
    for cname, cls in ABI_STRUCTS.items():      # the ctypes restatement of every argument struct against the library's own sizeof
        if lib.echr_abi_sizeof(cname.encode()) != C.sizeof(cls):
            raise EchrHipError('%s: ctypes layout is %d bytes, libechr_hip.so says %d -- rebuild the library (python __graft_entry__.py build)'
                               % (cname, C.sizeof(cls), lib.echr_abi_sizeof(cname.encode())))
    _lib = lib
    return lib


# Objects that keep host-side counts of work the device may have skipped (optimiser step counts): every site that can surface the
# asynchronous failure -62 goes through check(), which lets them re-read the device's own counts BEFORE the error propagates
# (persist_check_async has synchronised the device by then: the counts are final).
import weakref
ABORT_LISTENERS = weakref.WeakSet()


def check(rc, what=''):
    if rc != 0:
        msg = load().echr_last_error()
        msg = msg.decode() if msg else ''
        if rc == -62:
            for o in list(ABORT_LISTENERS):
                o._on_async_abort()
        raise EchrHipError('%s failed (rc=%d): %s' % (what, rc, msg))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """The current HIP stream as a void* (torch.cuda.current_stream() builds a Python Stream object per call: ~8 us; the raw getter 0.3)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32, name='tensor'):
    """Raw device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EchrHipError('%s must live on the GPU (got %s); echr_amd has no CPU path' % (name, t.device))
    if t.dtype != dtype:
        raise EchrHipError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise EchrHipError('%s must be contiguous' % name)
    return t.data_ptr()


def ptr3(ts, name):
    return (c_f * 3)(*[ptr(t, name=name) for t in ts])
