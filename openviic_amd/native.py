"""ctypes binding of ``libovc.so`` (C ABI in ``include/ovc.h``).

The library is the product: if it is missing or a call fails this module raises -- there is no
ATen or CPU fallback anywhere in ``openviic_amd``.  PyTorch only owns device memory and the
current HIP stream here; every pointer handed over is a raw ``data_ptr()``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_void_p

OVC_MAX_LAYERS = 8
OVC_MAX_LEVELS = 4
OVC_MAX_BEAM = 8
OVC_MAX_REGIONS = 1024
OVC_PROFILE_CLASSES = 4
ABI_VERSION = 7

_ERRORS = {-1: "OVC_EINVAL (bad argument / unsupported shape)", -2: "OVC_EWORKSPACE (workspace too small)",
           -3: "OVC_ELAUNCH (HIP launch failed)",
           -4: "OVC_EDEVICE (the library is bound to another device: one device per process)"}


class OvcError(RuntimeError):
    pass


class Lin(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("planes", c_void_p)]


class Norm(ctypes.Structure):
    _fields_ = [("g", c_void_p), ("b", c_void_p)]


class Mha(ctypes.Structure):
    _fields_ = [("q", Lin), ("k", Lin), ("v", Lin), ("o", Lin), ("ln", Norm), ("aoa_i", Lin), ("aoa_g", Lin),
                ("m_k", c_void_p), ("m_v", c_void_p)]


class Ffn(ctypes.Structure):
    _fields_ = [("fc1", Lin), ("fc2", Lin), ("ln", Norm)]


class EncLayer(ctypes.Structure):
    _fields_ = [("att", Mha), ("ffn", Ffn)]


class DecLayer(ctypes.Structure):
    _fields_ = [("self_att", Mha), ("cross_att", Mha), ("ffn", Ffn), ("alpha", Lin * OVC_MAX_LEVELS)]


class Model(ctypes.Structure):
    _fields_ = [
        ("abi", c_int32), ("enc_kind", c_int32), ("dec_kind", c_int32),
        ("d_feat", c_int32), ("d_model", c_int32), ("heads", c_int32), ("d_k", c_int32), ("d_v", c_int32),
        ("d_ff", c_int32), ("n_enc", c_int32), ("n_dec", c_int32), ("n_levels", c_int32), ("memory", c_int32),
        ("trig", c_int32), ("d_g", c_int32), ("vocab", c_int32), ("max_len", c_int32), ("pad_idx", c_int32),
        ("bos_idx", c_int32), ("eos_idx", c_int32), ("ln_eps", c_float),
        ("proj", Lin), ("enc_ln", Norm), ("fc_g_w", c_void_p), ("fc_g_b", c_void_p),
        ("enc", EncLayer * OVC_MAX_LAYERS), ("dec", DecLayer * OVC_MAX_LAYERS),
        ("word_emb", c_void_p), ("pos_emb", c_void_p), ("fc", c_void_p), ("fc_planes", c_void_p), ("tune_objective", c_int32), ("precision", c_int32),
    ]


ENC_PLAIN, ENC_MULTILEVEL, ENC_GEOMETRIC = 0, 1, 2
DEC_PLAIN, DEC_MESHED = 0, 1

# OVC_LIBRARY: load another build of the same ABI (A/B timing of kernel changes on one box)
LIBRARY_PATH = os.environ.get("OVC_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libovc.so")

# name -> (restype, argtypes); exactly the entry points declared in include/ovc.h
SIGNATURES = {
    "ovc_abi_version": (c_int, []),
    "ovc_build_info": (c_char_p, []),
    "ovc_bound_device": (c_int, []),
    "ovc_debug_rebind_device": (c_int, [c_int]),
    "ovc_linear": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ovc_layer_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                               c_void_p, c_int, c_int, c_void_p]),
    "ovc_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                              c_void_p, c_void_p]),
    "ovc_zero_row_mask": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ovc_region_position_encoding": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_float, c_void_p, c_void_p]),
    "ovc_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ovc_sigmoid_gate": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "ovc_gated_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p]),
    "ovc_log_softmax": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ovc_box_relation_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ovc_beam_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "ovc_workspace_bytes": (c_size_t, [POINTER(Model), c_int, c_int, c_int, c_int]),
    "ovc_encode": (c_int, [POINTER(Model), c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "ovc_beam_search": (c_int, [POINTER(Model), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "ovc_gemm_tune": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ovc_gemm_tune_calls": (c_long, []),
    "ovc_gemm_tuned_get": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "ovc_gemm_tuned_set": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "ovc_engine_gemm_shapes": (c_int, [POINTER(Model), c_int, c_int, c_int, POINTER(c_int32), c_int]),
    "ovc_graph_cache_drop_workspace": (c_int, [c_void_p]),
    "ovc_graph_cache_size": (c_int, []),
    "ovc_debug_force_gemm_tiling": (c_int, [c_int]),
    "ovc_debug_clear_tuning": (c_int, []),
    "ovc_debug_linear_tiling": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ovc_debug_vocab_select_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ovc_debug_vocab_select": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_size_t, c_void_p, c_void_p, c_void_p]),
    "ovc_split_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ovc_split_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ovc_debug_linear_planes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    "ovc_debug_repeat_linear": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ovc_beam_search_graph": (c_int, [POINTER(Model), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                      c_void_p, c_void_p, c_void_p]),
    "ovc_beam_search_early": (c_int, [POINTER(Model), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                      c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "ovc_graph_cache_clear": (c_int, []),
    "ovc_profile_enable": (c_int, [c_int]),
    "ovc_profile_read": (c_int, [c_int, c_int, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "ovc_profile_overhead_ms": (c_double, []),
    "ovc_profile_kernel_name": (c_char_p, [c_int]),
}

_lib = None


def load():
    """Load ``libovc.so`` (once).  Raises ``OvcError`` when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise OvcError("HIP library not built: {} is missing -- run `python -m openviic_amd.csrc.build` "
                       "(there is no CPU fallback)".format(LIBRARY_PATH))
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = restype, argtypes
    if lib.ovc_abi_version() != ABI_VERSION:
        raise OvcError("libovc.so ABI {} != binding ABI {}; rebuild".format(lib.ovc_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise OvcError("{} failed: {}".format(what, _ERRORS.get(status, status)))


def stream_handle():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
