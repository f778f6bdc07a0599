"""ctypes binding of the C-ABI library libmdd_hip.so (include/mdd_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product
path raises.  PyTorch only supplies device memory (`tensor.data_ptr()`) and the HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdd_hip.so")

DTYPE_F32, DTYPE_BF16, DTYPE_BF16X2, DTYPE_F32_BF16OPS = 0, 1, 2, 3
ABI_VERSION = 6     # include/mdd_hip.h MDD_ABI_VERSION


class MddConfig(C.Structure):
    _fields_ = [("variant", C.c_char_p), ("batch", C.c_int32), ("num_queries", C.c_int32),
                ("image_size", C.c_int32), ("d_txt", C.c_int32), ("syn_steps", C.c_int32),
                ("dtype", C.c_int32), ("keep_steps", C.c_int32)]


class MddIterArgs(C.Structure):
    _fields_ = [("image_syn", C.c_void_p), ("text_syn", C.c_void_p), ("lr_img", C.c_void_p),
                ("lr_txt", C.c_void_p), ("theta0_img", C.c_void_p), ("theta0_txt", C.c_void_p),
                ("target_img", C.c_void_p), ("target_txt", C.c_void_p), ("perms", C.c_void_p),
                ("drop_masks", C.c_void_p), ("syn_steps", C.c_int32),
                ("use_lr_as_scale", C.c_int32), ("logit_scale_const", C.c_float),
                ("grad_image_syn", C.c_void_p), ("grad_text_syn", C.c_void_p),
                ("grad_lr", C.c_void_p), ("losses", C.c_void_p)]


# every symbol include/mdd_hip.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "mdd_last_error": (C.c_char_p, []),
    "mdd_version": (_I, []),
    "mdd_engine_create": (_I, [C.POINTER(MddConfig), C.POINTER(_P)]),
    "mdd_engine_destroy": (None, [_P]),
    "mdd_engine_workspace_bytes": (_L, [_P]),
    "mdd_engine_bind_workspace": (_I, [_P, _P, _L, _P]),
    "mdd_engine_param_numel": (_L, [_P, _I]),
    "mdd_engine_param_count": (_I, [_P, _I]),
    "mdd_engine_param_info": (_I, [_P, _I, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_I),
                                   C.POINTER(_L)]),
    "mdd_engine_feature_dim": (_I, [_P]),
    "mdd_engine_find_buffer": (_I, [_P, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L),
                                    C.POINTER(_I)]),
    "mdd_img_forward": (_I, [_P, _I, _P, _P, _P, _P, _P]),
    "mdd_img_backward": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "mdd_img_tangent_forward": (_I, [_P, _I, _P, _P, _P, _P]),
    "mdd_img_tangent_backward": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "mdd_txt_forward": (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "mdd_txt_backward": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _I, _P]),
    "mdd_txt_tangent_forward": (_I, [_P, _I, _P, _P, _P, _P]),
    "mdd_txt_tangent_backward": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "mdd_contrastive": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P]),
    "mdd_contrastive_tangent": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
    "mdd_flat_axpy": (_I, [_P, _P, _P, _P, _F, _L, _P]),
    "mdd_flat_sqdist": (_I, [_P, _P, _P, _L, _P]),
    "mdd_flat_sgd_momentum": (_I, [_P, _P, _P, _F, _F, _I, _L, _P]),
    "mdd_flat_sgd_momentum_guarded": (_I, [_P, _P, _P, _F, _F, _I, _L, _P, _P]),
    "mdd_engine_profile": (_I, [_P, _I]),
    "mdd_engine_set_pass_precision": (_I, [_P, _I, _I, _I, _I]),
    "mdd_engine_profile_read": (_I, [_P, _I, C.POINTER(C.c_double)]),
    "mdd_engine_profile_dump": (_I, [_P, C.c_char_p]),
    "mdd_unrolled_match": (_I, [_P, C.POINTER(MddIterArgs), _P]),
    "mdd_op_conv2d": (_I, [_I] * 11 + [_P, _P, _P, _P, _P]),
    "mdd_op_conv2d_wgrad": (_I, [_I] * 10 + [_P, _P, _P, _P, _P]),
    "mdd_set_pipe_kernels": (_I, [_I]),
    "mdd_op_conv2d_wgrad2": (_I, [_I] * 10 + [_P] * 7 + [C.c_longlong, _P]),
    "mdd_op_contrastive_workspace_floats": (_L, [_I, _I]),
    "mdd_op_contrastive": (_I, [_I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P]),
    "mdd_retrieval_ranks": (_I, [_P] * 5 + [_I] * 3 + [C.c_float] + [_P] * 5),
    "mdd_nearest_neighbor": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mdd_comm_unique_id": (_I, [_P]),
    "mdd_comm_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "mdd_comm_destroy": (None, [_P]),
    "mdd_comm_world": (_I, [_P]),
    "mdd_allreduce_syn_grads": (_I, [_P, _P, _L, _I, _P]),
    "mdd_op_layernorm": (_I, [_I, _I, _I, _F] + [_P] * 9),
    "mdd_op_add_layernorm": (_I, [_I, _I, _I, _F] + [_P] * 13),
    "mdd_op_layernorm_bwd": (_I, [_I, _I, _I, _F] + [_P] * 15),
    "mdd_op_gelu": (_I, [_I, _L] + [_P] * 5),
    "mdd_op_gelu_bwd": (_I, [_I, _L] + [_P] * 7),
    "mdd_op_attention": (_I, [_I, _I, _I, _I, _F] + [_P] * 12 + [_I, _I, _I, _P]),
    "mdd_op_softmax": (_I, [_I, _L, _I, _I, _F] + [_P] * 5),
    "mdd_op_softmax_bwd": (_I, [_I, _L, _I, _I, _F] + [_P] * 7),
    "mdd_op_bgemm": (_I, [_I, _I, _I] + [_P] * 8),
}


class MddBgemmDesc(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("outer", C.c_int32), ("inner", C.c_int32),
                ("a_row", C.c_int64), ("a_col", C.c_int64), ("b_row", C.c_int64), ("b_col", C.c_int64),
                ("c_row", C.c_int64), ("c_col", C.c_int64),
                ("a_outer", C.c_int64), ("a_inner", C.c_int64), ("b_outer", C.c_int64), ("b_inner", C.c_int64),
                ("c_outer", C.c_int64), ("c_inner", C.c_int64), ("alpha", C.c_float)]


COMM_ID_BYTES = 128   # include/mdd_hip.h MDD_COMM_ID_BYTES

_lib = None


def load(variant=None):
    """Load libmdd_hip.so and bind every declared symbol.  Raises (never falls back).

    The product path always loads the in-tree library: no environment variable can swap it (MDD_HIP_LIB is
    ignored; distill.py / buffer.py record any MDD_* variable they see).  Kernel-variant experiments under tools/
    pass the variant library's path explicitly (`load(variant=...)`, or set `_lib.LIB_PATH` before the first load).
    """
    global _lib
    if _lib is not None:
        return _lib
    path = variant if variant is not None else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "libmdd_hip.so is not built (%s). Run `python -m multimodal_dataset_distillation_amd."
            "build_ext` (needs hipcc, gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.mdd_version() != ABI_VERSION:      # MddConfig / MddIterArgs layouts below belong to this version
        raise RuntimeError("libmdd_hip.so ABI version %d, this binding expects %d: rebuild (build_ext)"
                           % (lib.mdd_version(), ABI_VERSION))
    _lib = lib
    return lib


def stray_env():
    """MDD_* variables present in the environment (none of them changes what the product library does)."""
    return sorted(k for k in os.environ if k.startswith("MDD_"))


def check(rc):
    if rc != 0:
        msg = load().mdd_last_error().decode("utf-8", "replace")
        raise RuntimeError("mdd_hip error %d: %s" % (rc, msg))
