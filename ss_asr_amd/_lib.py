"""ctypes binding of libssasr_hip.so (the C ABI declared in include/ssasr.h).

There is no fallback: if the shared object is missing or fails to load, the
import error says how to build it, and every op of this package fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SSASR_LIB') or os.path.join(_HERE, 'libssasr_hip.so')   # SSASR_LIB: A/B builds
ABI_VERSION = 14

P = C.c_void_p
I64 = C.c_int64
F32 = C.c_float
I32 = C.c_int


class Decoder(C.Structure):
    """struct ssasr_decoder (include/ssasr.h) -- field order is the ABI."""
    _fields_ = (
        [(n, I64) for n in ('B', 'T', 'E', 'A', 'D', 'V', 'U')] +
        [('feat', P), ('comp', P), ('enc_len', P), ('teacher', P), ('teacher_ld', I64),
         ('step_mode', P), ('uniforms', P),
         ('w_phi', P),
         ('w_ih1', P), ('w_hh1', P), ('b_ih1', P), ('b_hh1', P),
         ('w_ih2', P), ('w_hh2', P), ('b_ih2', P), ('b_hh2', P),
         ('embed', P), ('w_ct', P), ('b_ct', P),
         ('logits', P), ('att', P),
         ('w_phi_t', P), ('q', P), ('ctx', P), ('emb_in', P), ('chars', P),
         ('gates1', P), ('c1', P), ('h1', P), ('gates2', P), ('c2', P), ('h2', P),
         ('ws_hx1', P), ('ws_hx2', P), ('ws_qx', P), ('ws_modes', P), ('ws_sync', P),
         ('modes_ready', C.c_int32), ('ws_armed', C.c_int32), ('ws_attn', P), ('ws_attn_phase', C.c_int32),
         ('ws_part', P)])


class DecoderGrads(C.Structure):
    """struct ssasr_decoder_grads (include/ssasr.h)."""
    _fields_ = [(n, P) for n in (
        'dlogits', 'dfeat', 'dcomp', 'dw_phi', 'dw_ih1', 'dw_hh1', 'db1', 'dw_ih2', 'dw_hh2',
        'db2', 'dembed', 'dw_ct', 'db_ct', 'ws_t_ih1', 'ws_t_hh1', 'ws_t_ih2', 'ws_t_hh2',
        'ws_dh2', 'ws_dctx', 'ws_de', 'ws_dqpre', 'ws_dc', 'ws_demb', 'ws_gx', 'ws_sync', 'ws_chain',
        'db1_2', 'db2_2')] + [('defer_wgrad', C.c_int32), ('ws_armed', C.c_int32)]


SIGNATURES = {
    'ssasr_abi_version': (I32, []),
    'ssasr_set_option': (I32, [C.c_char_p, I32]),
    'ssasr_get_option': (I32, [C.c_char_p, C.POINTER(C.c_int)]),
    'ssasr_events_create': (I32, [C.POINTER(P)]),
    'ssasr_events_destroy': (I32, [P]),
    'ssasr_gemm_f32': (I32, [I32, I32, I64, I64, I64, F32, P, I64, P, I64, F32, P, I64, P, I32,
                             I64, I64, I64, I64, I32, P]),
    'ssasr_bilstm_fwd': (I32, [P, I64, I64, I64, I64, I64, I64, P] + [P] * 8 +
                         [P, I64, I64, P, P, P, P, P, I32, P, P]),
    'ssasr_bilstm_tsave_floats': (I64, [I64, I64, I64]),
    'ssasr_bilstm_fwd_hx_floats': (I64, [I64, I64, I64]),
    'ssasr_bilstm_bwd': (I32, [P, I64, I64, P, I64, I64, I64, I64, I64, I64, P] + [P] * 4 +
                         [P, P, P, P, I64, I64] + [P] * 6 + [P, P, P, P, I32, P, P]),
    'ssasr_bilstm_wgrad': (I32, [P, P, I64, I64, P, I64, I64, I64, I64] + [P] * 8 + [I32, P]),
    'ssasr_bilstm_bwd_overlapped': (I32, [P, I64, I64, P, I64, I64, I64, I64, I64, I64, P] + [P] * 4 +
                                    [P, P, P, P, I64, I64] + [P] * 8 + [P, P, P, P, I32, P, I32, P, P, P]),
    'ssasr_lstm_cell_fwd': (I32, [P, I64, I64, P, I64, I64, P, P, P, P, P, P, I64, I64, P, P, P, P]),
    'ssasr_lstm_cell_bwd': (I32, [P, P, P, P, P, I64, I64, P, P, P]),
    'ssasr_attn_precompute_fwd': (I32, [P, P, P, I64, I64, I64, P, P]),
    'ssasr_attn_precompute_bwd': (I32, [P, P, P, P, I64, I64, I64, P, P, P, P]),
    'ssasr_attn_precompute_wgrad': (I32, [P, P, I64, I64, I64, P, P, I32, P]),
    'ssasr_attn_step_ws_floats': (I64, [I64, I64, I64, I64]),
    'ssasr_attn_step_fwd': (I32, [P, P, P, P, P, I64, I64, I64, I64, I64, P, P, P, P, I32, P, P]),
    'ssasr_attn_step_bwd': (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, P, P, P]),
    'ssasr_decoder_fwd_part_floats': (I64, [I64] * 6),
    'ssasr_decoder_fwd': (I32, [C.POINTER(Decoder), P]),
    'ssasr_decoder_bwd': (I32, [C.POINTER(Decoder), C.POINTER(DecoderGrads), P]),
    'ssasr_decoder_wgrad': (I32, [C.POINTER(Decoder), C.POINTER(DecoderGrads), I32, P]),
    'ssasr_ce_loss_fwd': (I32, [P, P, I64, I64, I64, I64, I64, P, P, P]),
    'ssasr_ce_loss_bwd': (I32, [P, P, I64, P, P, I64, I64, I64, P, P]),
    'ssasr_ctc_ws_floats': (I64, [I64, I64, I64, I64]),
    'ssasr_ctc_loss_fwd': (I32, [P, P, P, I64, P, I64, I64, I64, I64, I32, P, P, P]),
    'ssasr_ctc_loss_bwd': (I32, [P, P, P, I64, P, I64, I64, I64, I64, I32, P, P, P, P, P]),
    'ssasr_clip_adadelta_ws': (I64, [I64]),
    'ssasr_clip_adadelta': (I32, [P, P, P, P, I64, F32, F32, F32, F32, F32, P, P, I32, P]),
    'ssasr_adam_ws': (I64, [I64]),
    'ssasr_adam_prepare': (I32, [P, I64, F32, F32, F32, F32, F32, P, P, P, P]),
    'ssasr_adam_update': (I32, [P, P, P, P, I64, P, I32, F32, F32, F32, F32, P, I32, P]),
    'ssasr_linear_fwd': (I32, [P, I64, P, P, P, I64, I64, I64, I32, P]),
    'ssasr_linear_bwd': (I32, [P, P, P, I64, P, P, I64, P, P, I64, I64, I64, I32, P]),
    'ssasr_act_bwd': (I32, [I32, P, P, P, I64, P]),
    'ssasr_bce_fwd': (I32, [P, I64, F32, P, P]),
    'ssasr_bce_bwd': (I32, [P, I64, F32, P, P, P]),
    'ssasr_conv2d_ws_floats': (I64, [I64] * 7),
    'ssasr_conv2d_fwd': (I32, [P, P, P] + [I64] * 7 + [P, P]),
    'ssasr_conv2d_bwd': (I32, [P, I32, P, P, P, P] + [I64] * 7 + [P, P]),
    'ssasr_bn_ws_floats': (I64, [I64]),
    'ssasr_bn_stats': (I32, [P, I64, I64, P, P, P, P, F32, F32, I32, P, P, P]),
    'ssasr_pool_ws_floats': (I64, [I64] * 6),
    'ssasr_bn_relu_pool_fwd': (I32, [P, P] + [I64] * 6 + [P, P, P, P]),
    'ssasr_bn_relu_pool_bwd': (I32, [P, P, P, P, P, P] + [I64] * 8 + [P, P, P, P, P]),
    'ssasr_sae_concat_fwd': (I32, [P, P, I64, I64, I64, I64, P, P]),
    'ssasr_sae_concat_bwd': (I32, [P, I64, I64, I64, I64, P, P, P]),
    'ssasr_smooth_l1_ws_floats': (I64, []),
    'ssasr_smooth_l1_fwd': (I32, [P, P] + [I64] * 5 + [P, P, P]),
    'ssasr_smooth_l1_bwd': (I32, [P, P] + [I64] * 5 + [P, P, P]),
    'ssasr_bilstm_bwd_gx_floats': (I64, [I64, I64, I64]),
    'ssasr_bilstm_bwd_ring_floats': (I64, [I64, I64, I64, I64]),
    'ssasr_decoder_bwd_chain_floats': (I64, [I64] * 6),
    'ssasr_frame_lengths': (I32, [P, I64, I64, I64, P, P]),
    'ssasr_gather_batch': (I32, [P, P, P, I64, I64, I64, P, P]),
    'ssasr_logmel_frames': (I64, [I64, I64, I64]),
    'ssasr_logmel_batch_rows': (I64, [I64, I64, I64]),
    'ssasr_logmel_batch': (I32, [P, P, I64, I64, I64, I64, I64, I64, P, P, P, P, P, P]),
    'ssasr_logmel': (I32, [P, I64, I64, I64, I64, P, P, P, P, P, P, P, P]),
}

_lib = None


def load():
    """Loads the shared object once; raises if it is missing (no CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 / HSA runtime.  It must be the copy
    # this process binds: loading ours first would pull in /opt/rocm's runtime
    # and the two stacks then disagree about the device (hipErrorNoDevice).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'ss_asr_amd: %s not found. Build it with `python -m ss_asr_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no fallback path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ssasr_abi_version() != ABI_VERSION:
        raise ImportError('ss_asr_amd: ABI version mismatch, rebuild libssasr_hip.so')
    _lib = lib
    return lib


def set_option(name, value):
    """Diagnostic switch of the library (include/ssasr.h, ssasr_set_option); returns the old value."""
    lib = load()
    old = C.c_int(0)
    if lib.ssasr_get_option(name.encode(), C.byref(old)) != 0:
        raise KeyError(name)
    lib.ssasr_set_option(name.encode(), int(value))
    return old.value


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError('%s: invalid argument (code %d)' % (what, rc))
    raise RuntimeError('%s: HIP error %d' % (what, rc))
