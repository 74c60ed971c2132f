"""
ctypes binding of libgcnpt.so (include/gcnpt.h).  There is no CPU fallback: if the HIP library has
not been built this module raises, and every op that needs it fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCNPT_LIB") or os.path.join(_HERE, "csrc", "libgcnpt.so")   # GCNPT_LIB: diagnostic builds

F32, BF16 = 0, 1
ABI_VERSION = 7            # GCNPT_ABI_VERSION of the include/gcnpt.h this binding was written against
OPT_DETERMINISTIC, OPT_FOUR_WAVES, OPT_SIDE_TILES, OPT_COL_SPLIT = 0, 1, 2, 3      # gcnpt_set_option keys (include/gcnpt.h)
OK, E_INVALID, E_PRUNE_NEGATIVE, E_NO_SUBJECT, E_NO_LCA, E_CYCLE, E_BAD_HEAD, E_ASSERT, E_CAPACITY, E_HIP, E_UNSUPPORTED, E_LENGTH = \
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11

# every symbol include/gcnpt.h declares: (restype, argtypes)
_p, _i, _f, _u64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_size_t


class Step(ctypes.Structure):
    """gcnpt_step_t of include/gcnpt.h: a whole step's arguments, marshalled once (gcnpt_layers_step)."""
    _fields_ = [("n_layers", _i), ("B", _i), ("T", _i), ("compute_dtype", _i), ("parts", _i), ("gy_is_dz", _i),
                ("W", _p * 8), ("bias", _p * 8), ("Din", _i * 8), ("H", _i * 8), ("w_fwd", _p * 8), ("w_bwd", _p * 8),
                ("row_ptr", _p), ("col_idx", _p), ("ell", _p), ("deg_ell", _p), ("rowT_ptr", _p), ("colT_idx", _p), ("ellT", _p), ("ell_bwd", _p),
                ("x", _p), ("x_dtype", _i), ("out", _p * 8), ("out_dtype", _i * 8), ("drop_p", _f * 8), ("seed", _u64 * 8), ("seed_dev", _p),
                ("s_frag", _p * 8),
                ("gy", _p), ("dh", _p * 8), ("dh_dtype", _i * 8), ("scale", _f * 8), ("z_frag", _p * 8), ("dW", _p * 8), ("db", _p * 8)]


STEP_PACK, STEP_FWD, STEP_BWD = 1, 2, 4         # gcnpt_step_t.parts
SIGNATURES = {
    "gcnpt_abi_version": (_i, []),
    "gcnpt_last_error": (ctypes.c_char_p, []),
    "gcnpt_prune_to_csr": (_i, [_p] * 7 + [_i] * 4 + [_p] * 9),
    "gcnpt_prune_to_csr_pack": (_i, [_p] * 7 + [_i] * 4 + [_p] * 9 + [_i, _p, _p, _p, _i, _p, _p]),
    "gcnpt_adj_to_csr": (_i, [_p, _p, _i, _i, _i] + [_p] * 9),
    "gcnpt_csr_to_adj": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "gcnpt_packed_bytes": (_sz, [_i, _i, _i]),
    "gcnpt_pack_weights": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "gcnpt_pack_weights_multi": (_i, [_p, _i, _p, _p, _p, _i, _p, _p]),
    "gcnpt_frag_bytes": (_sz, [_i, _i, _i]),
    "gcnpt_layer_fwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _f, _u64, _p, _p]),
    "gcnpt_layer_bwd_data": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _f, _p, _p, _p, _p, _f, _i]),
    "gcnpt_layer_bwd_data_wgrad": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _f, _p, _p, _p, _p, _f, _i] + [_p, _p, _i, _i, _p, _p]),
    "gcnpt_layers_bwd_range": (_i, [_p, _i] + [_p] * 8 + [_i, _i] + [_p] * 4 + [_i] + [_p] * 5 + [_i, _i, _i]),
    "gcnpt_set_option": (_i, [_i, _i]),
    "gcnpt_get_option": (_i, [_i]),
    "gcnpt_last_launch": (_i, [_p, _p, _p, _p]),
    "gcnpt_launch_empty": (_i, [_p, _i, _i, _i, _i]),
    "gcnpt_launch_empty_seq": (_i, [_p, _i, _p, _p, _p, _p]),
    "gcnpt_layer_bwd_weight": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _i]),
    "gcnpt_layer_bwd_weight_multi": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _i]),
    "gcnpt_layers_fwd": (_i, [_p, _i, _p, _i] + [_p] * 6 + [_i, _i] + [_p] * 4 + [_i] + [_p] * 4),
    "gcnpt_layers_step": (_i, [_p, ctypes.POINTER(Step)]),
    "gcnpt_layers_bwd": (_i, [_p, _i] + [_p] * 8 + [_i, _i] + [_p] * 4 + [_i] + [_p] * 5),
    "gcnpt_prune_to_csr_packed": (_i, [_p] * 7 + [_i, _i, _i] + [_p] * 10 + [_i, _i] + [_p] * 4 + [_i, _p, _p, _p, _i, _p, _p]),
    "gcnpt_gather_trees_packed": (_i, [_p] * 11 + [_i, _i, _i, _p, _i, _i] + [_p] * 10 + [_i, _i] + [_p] * 3 + [_i, _p, _p, _p, _i, _p, _p]),
    "gcnpt_sgd_clip_update": (_i, [_p, _p, _p, ctypes.c_longlong, _f, _f, _f, _p, _p]),
    "gcnpt_pack_trees": (_i, [_p] * 10 + [_i, _i, _i] + [_p] * 10 + [_i, _i, _p]),
    "gcnpt_pack_rows": (_i, [_p, _p, _i, _p, _i, _i, _i, _p]),
    "gcnpt_unpack_rows": (_i, [_p, _p, _i, _p, _i, _i, _i, _p]),
    "gcnpt_full_agg_fwd": (_i, [_p] * 10 + [_i, _i, _i, _i, _p, _f, _u64, _p]),
    "gcnpt_full_agg_bwd": (_i, [_p] * 9 + [_i, _i, _i, _i, _f, _p, _p, _p]),
    "gcnpt_pool3_fwd": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "gcnpt_pool3_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i]),
    "gcnpt_pool3_bwd_dz": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _p, _i]),
    "gcnpt_layers_bwd_dz": (_i, [_p, _i] + [_p] * 8 + [_i, _i] + [_p] * 4 + [_i] + [_p] * 5),
    "gcnpt_diag_layer_fwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _f, _u64, _p]),
    "gcnpt_diag_layer_bwd": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _f]),
    "gcnpt_gather_trees": (_i, [_p] * 11 + [_i, _i, _i, _p, _i, _i, _i] + [_p] * 9),
    "gcnpt_gather_trees_pack": (_i, [_p] * 11 + [_i, _i, _i, _p, _i, _i, _i] + [_p] * 9 + [_i, _p, _p, _p, _i, _p, _p]),
    "gcnpt_compact_trees": (_i, [_p] * 10 + [_i] * 5 + [_p] * 11),
    "gcnpt_bilinear_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "gcnpt_bilinear_supported": (_i, [_i, _i, _i, _i]),
    "gcnpt_bilinear_planes": (_i, [_i, _i, _i, _i, _i]),
    "gcnpt_bilinear_pack": (_i, [_p, _p, _i, _i, _i, _p, _i, _i]),
    "gcnpt_rows_image_bytes": (_sz, [_i, _i, _i]),
    "gcnpt_rows_pack": (_i, [_p, _p, _i, _i, _p, _i]),
    "gcnpt_bilinear_bwd_w": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i]),
    "gcnpt_bilinear_de_planes": (_i, [_i, _i, _i, _i, _i]),
    "gcnpt_bilinear_bwd_e": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i]),
    "gcnpt_bilinear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i]),
}


class GcnptError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libgcnpt error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """The loaded library.  Raises ImportError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C %s` (or python -c 'import __graft_entry__ as g; g.build()'). "
                "This package has no CPU fallback." % (LIB_PATH, os.path.dirname(LIB_PATH)))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = header and library disagree
            fn.restype, fn.argtypes = res, args
        if handle.gcnpt_abi_version() != ABI_VERSION:
            raise ImportError("libgcnpt.so has ABI version %d, this binding needs %d" % (handle.gcnpt_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def set_option(key, value):
    """Process-wide option of the library (include/gcnpt.h, GCNPT_OPT_*); returns the previous value."""
    old = lib().gcnpt_get_option(key)
    check(lib().gcnpt_set_option(key, int(value)))
    return old


def check(rc):
    if rc != 0:
        raise GcnptError(rc, lib().gcnpt_last_error().decode("utf-8", "replace"))


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise TypeError("gcnpt supports float32 and bfloat16 tensors, got %s" % torch_dtype)


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("gcnpt ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)
    return t


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and live on a GPU."""
    if t is None:
        return None
    require_gpu(t)
    if not t.is_contiguous():
        raise RuntimeError("gcnpt ops need contiguous tensors")
    return t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers (None -> NULL) for the entry points that take one pointer per layer."""
    return (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
