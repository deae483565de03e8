"""ctypes binding of libivg.so (include/ivg.h).  The HIP library is the product: there is no CPU or
PyTorch fallback -- importing this module without a built library raises immediately."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libivg.so")

IVG_F32, IVG_BF16, IVG_F32X3 = 0, 1, 2
IVG_K_IGEMM_BF16, IVG_K_IGEMM_F32, IVG_K_CONV3X3_BF16, IVG_K_CONV3X3_F32, IVG_K_DECODE_ATTN, IVG_K_DECODE_GEMM = 0, 1, 2, 3, 4, 5
# igemm epilogue flags (csrc/igemm.h)
IG_BIAS_N, IG_BIAS_M, IG_RESIDUAL, IG_SILU, IG_GLU, IG_OUT_F32 = 1, 2, 4, 8, 16, 32


class IvgTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class IvgConfig(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * 8), ("layers_per_block", C.c_int32),
        ("latent_channels", C.c_int32), ("vq_embed_dim", C.c_int32), ("num_vq_embeddings", C.c_int32),
        ("num_dyn_embeddings", C.c_int32), ("norm_num_groups", C.c_int32), ("mid_block_add_attention", C.c_int32),
        ("context_length", C.c_int32), ("max_att_resolution", C.c_int32), ("resolution", C.c_int32),
        ("patch_size", C.c_int32),
        ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("vocab_size", C.c_int32), ("max_position_embeddings", C.c_int32),
        ("rms_norm_eps", C.c_float), ("action_dim", C.c_int32), ("reward_head", C.c_int32),
        ("encode_dtype", C.c_int32), ("decode_dtype", C.c_int32), ("llm_dtype", C.c_int32),
        ("max_batch", C.c_int32), ("max_frames", C.c_int32), ("max_seq", C.c_int32),
        ("decode_lds_kb", C.c_int32),
    ]


class IvgProfileStats(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("total_flops", C.c_double),
                ("total_bytes", C.c_double)]


class IvgIgemmArgs(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("W", C.c_void_p), ("Y", C.c_void_p), ("R", C.c_void_p), ("bias", C.c_void_p),
        ("Nimg", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("ldx", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
        ("pad", C.c_int32), ("ups", C.c_int32), ("N", C.c_int32), ("ldw", C.c_int32),
        ("c_img", C.c_int64), ("c_pix", C.c_int64), ("c_ch", C.c_int64), ("c_grp_stride", C.c_int64),
        ("c_grp", C.c_int32), ("flags", C.c_int32), ("alpha", C.c_float),
        ("nb0", C.c_int32), ("nb1", C.c_int32), ("nb2", C.c_int32),
        ("sa", C.c_int64 * 3), ("sw", C.c_int64 * 3), ("sy", C.c_int64 * 3),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "ivg_version": (C.c_char_p, []),
    "ivg_last_error": (C.c_char_p, [C.c_void_p]),
    "ivg_create": (C.c_int, [C.POINTER(IvgConfig), C.POINTER(IvgTensor), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ivg_destroy": (None, [C.c_void_p]),
    "ivg_reload_switches": (None, []),
    "ivg_set_temperature": (C.c_int, [C.c_void_p, C.c_float]),
    "ivg_set_decode_lds_kb": (C.c_int, [C.c_void_p, C.c_int]),
    "ivg_set_context_length": (C.c_int, [C.c_void_p, C.c_int]),
    "ivg_tokenize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_encode_context": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "ivg_detokenize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ivg_detokenize_shared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ivg_detokenize_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ivg_set_output_clamp": (C.c_int, [C.c_void_p, C.c_int]),
    "ivg_cache_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ivg_cache_destroy": (None, [C.c_void_p, C.c_void_p]),
    "ivg_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_generate_shared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_generate_forced_sdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "ivg_generate_continue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ivg_action_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ivg_generate_embeds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "ivg_reward_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ivg_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ivg_eval_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_action_recon_sqerr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "ivg_ingest_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ivg_frame_metrics_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ivg_frame_metrics": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ivg_profile_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ivg_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(IvgProfileStats)]),
    "ivg_profile_attn_fit": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ivg_profile_gemm_kinds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ivg_op_igemm": (C.c_int, [C.POINTER(IvgIgemmArgs), C.c_int, C.c_void_p]),
    "ivg_op_conv_gn": (C.c_int, [C.POINTER(IvgIgemmArgs), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                 C.c_void_p]),
    "ivg_op_gn_conv": (C.c_int, [C.POINTER(IvgIgemmArgs), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "ivg_op_conv_x3": (C.c_int, [C.POINTER(IvgIgemmArgs), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "ivg_op_xattn": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]),
    "ivg_op_conv_subpixel": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ivg_op_shared_decode_attn": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 9 + [C.c_void_p]),
    "ivg_op_kv24_pack": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "ivg_op_decode_attn24": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "ivg_op_skinny": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "ivg_op_skinny_policy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]),
    "ivg_op_groupnorm": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "ivg_op_softmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 6 + [C.c_void_p]),
    "ivg_op_vq_argmin": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "ivg_op_add_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "ivg_op_conv_in": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "ivg_op_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ivg_debug_counter": (C.c_int64, [C.c_char_p]),
}

_lib = None


def load():
    """Load libivg.so once; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built (make -C ivideogpt_amd/csrc). "
            "ivideogpt_amd has no CPU / PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def reload_switches():
    """Publish the current IVG_* environment variables to the library (csrc/switches.h): read at load and at every ivg_create;
    a test that flips one between two op-level calls calls this."""
    load().ivg_reload_switches()


def last_error(handle=None):
    msg = load().ivg_last_error(handle)
    return msg.decode() if msg else ""


class IvgError(RuntimeError):
    pass


def check(rc, handle=None, what=""):
    if rc != 0:
        msg = f"{what}: libivg error {rc}: {last_error(handle)}"
        if rc == -1:
            raise AssertionError(msg)  # the reference raises AssertionError on shape / context-length mismatches
        raise IvgError(msg)
