"""ctypes binding of libbts_amd.so (the C ABI declared in include/bts_amd.h).

The library must share PyTorch's HIP runtime so that streams and device pointers are
interchangeable: ``import torch`` first (it loads its bundled ``libamdhip64.so``, SONAME
``libamdhip64.so.7``), then ``dlopen`` ours, whose DT_NEEDED entry resolves to the already
loaded runtime.  There is NO fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported before the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbts_amd.so")

BTS_F32, BTS_BF16 = 0, 1
ACT_NONE, ACT_ELU, ACT_SIGMOID, ACT_RELU = 0, 1, 2, 3
MAX_SEG, MAX_TAP, BN_MAX_SEG = 6, 16, 6
ERRORS = {-1: "BTS_ERR_ARG", -2: "BTS_ERR_LAUNCH", -3: "BTS_ERR_UNSUPPORTED"}


class BtsAmdError(RuntimeError):
    """code: the library's status (BTS_ERR_*) when the error comes from an entry point, else None."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_UNSUPPORTED = -3


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("stride", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("N", C.c_int32), ("Hg", C.c_int32), ("Wg", C.c_int32),
        ("nseg", C.c_int32), ("seg", Seg * MAX_SEG),
        ("Hx", C.c_int32), ("Wx", C.c_int32), ("isc", C.c_int32),
        ("nphase", C.c_int32), ("T", C.c_int32),
        ("dy", C.c_int16 * MAX_TAP), ("dx", C.c_int16 * MAX_TAP),
        ("ioy", C.c_int16 * MAX_TAP), ("iox", C.c_int16 * MAX_TAP),
        ("w", C.c_void_p), ("Cout", C.c_int32),
        ("y", C.c_void_p), ("y_dtype", C.c_int32), ("y_stride", C.c_int32),
        ("Hy", C.c_int32), ("Wy", C.c_int32), ("osc", C.c_int32), ("act", C.c_int32),
        ("out_scale", C.c_float), ("out_scale_n", C.c_void_p), ("accumulate", C.c_int32),
        ("fold_elu_y", C.c_void_p), ("fold_elu_stride", C.c_int32),
        ("w2", C.c_void_p), ("y2", C.c_void_p), ("Cout2", C.c_int32), ("y2_stride", C.c_int32), ("accumulate2", C.c_int32),
        ("w_frag", C.c_int32),
        ("stats_ws", C.c_void_p),
    ]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("cmap", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KK", C.c_int32), ("mode", C.c_int32),
                ("R", C.c_int32), ("K", C.c_int32), ("T", C.c_int32),
                ("tapmask", C.c_uint16 * MAX_TAP), ("first_block", C.c_int32), ("layout", C.c_int32), ("Tp", C.c_int32)]


class AugParams(C.Structure):
    _fields_ = [("crop_x", C.c_int32), ("crop_y", C.c_int32), ("flip", C.c_int32), ("augment", C.c_int32),
                ("gamma", C.c_float), ("brightness", C.c_float), ("color", C.c_double * 3)]


class UnpackJob(C.Structure):
    _fields_ = [("dwp_off", C.c_int64), ("gw_off", C.c_int64), ("kinv", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KK", C.c_int32), ("K", C.c_int32), ("T", C.c_int32),
                ("tapmask", C.c_uint16 * MAX_TAP), ("first_block", C.c_int32)]


class BnSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dx", C.c_void_p), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("C", C.c_int32), ("x_stride", C.c_int32), ("dx_stride", C.c_int32), ("accumulate", C.c_int32)]


class BnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("nseg", C.c_int32), ("M", C.c_int64), ("seg", BnSeg * BN_MAX_SEG),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("eps", C.c_float), ("momentum", C.c_float), ("relu", C.c_int32), ("use_batch_stats", C.c_int32),
                ("y", C.c_void_p), ("y_stride", C.c_int32), ("elu_x", C.c_int32), ("y2", C.c_void_p), ("y2_stride", C.c_int32)]


BN_MAX_MULTI = 4
BN_MULTI_TENSORS = 3


class BnContrib(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("dy_stride", C.c_int32), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("dbeta", C.c_void_p), ("dgamma", C.c_void_p)]


class BnMultiTensor(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int32), ("C", C.c_int32), ("mean", C.c_void_p), ("var", C.c_void_p),
                ("dx", C.c_void_p), ("dx_stride", C.c_int32), ("accumulate", C.c_int32), ("c", BnContrib * BN_MAX_MULTI)]


class BnMultiDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("nt", C.c_int32), ("relu", C.c_int32), ("M", C.c_int64),
                ("eps", C.c_float), ("use_batch_stats", C.c_int32), ("t", BnMultiTensor * BN_MULTI_TENSORS)]


_i, _l, _f, _p = C.c_int, C.c_long, C.c_float, C.c_void_p

# name -> argtypes (restype is int unless listed in _LONG_RET); mirrors include/bts_amd.h exactly
SIGNATURES = {
    "bts_abi_version": [],
    "bts_current_device": [],
    "bts_lpg_fwd": [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    "bts_lpg_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "bts_lpg_fwd_multi": [_i, _p, _p, _p, _p, _p, _p, _p, _p],
    "bts_lpg_bwd_multi": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "bts_lpg_head_fwd": [_p, _i, _p, _p, _i, _i, _i, _i, _f, _p],
    "bts_lpg_head_bwd": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "bts_plane_fwd": [_p, _i, _p, _l, _f, _p],
    "bts_plane_bwd": [_p, _i, _p, _p, _i, _i, _i, _l, _f, _p],
    "bts_lpg_chain_fwd": [_p, _i, _i, _i, _i, _p, _i, _p, _l, _i, _i, _i, _f, _p],
    "bts_lpg_chain_bwd": [_p, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _p, _p, _i, _l, _i, _i, _i, _f, _p],
    "bts_pack_maps": [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "bts_unpack_maps": [_p, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "bts_eval_workspace_bytes": [_i],
    "bts_eval_errors": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i, _p, _p, _p, _p],
    "bts_depth_to_u16": [_p, _p, _l, _f, _p],
    "bts_preprocess_train": [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p],
    "bts_silog_workspace_bytes": [_l],
    "bts_silog_fwd": [_p, _p, _p, _f, _l, _f, _p, _p, _p, _p],
    "bts_silog_bwd": [_p, _p, _p, _f, _l, _f, _p, _p, _p, _p, _p],
    "bts_conv_fwd": [C.POINTER(ConvDesc), _p],
    "bts_conv_fwd_stats_rows": [C.POINTER(ConvDesc), C.POINTER(C.c_int)],
    "bts_conv_wgrad": [C.POINTER(ConvDesc), _p, _i, _p, _p],
    "bts_conv_wgrad_group": [_p, _p, _p, _p, _i, _p],
    "bts_conv3x3_c1_fwd": [_p, _i, _i, _i, _p, _p, _i, _i, _i, _f, _p, _p],
    "bts_conv3x3_c1_dgrad": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _f, _p, _p],
    "bts_conv3x3_c1_wgrad": [_p, _p, _p, _i, _i, _i, _p, _i, _i, _i, _i, _f, _p, _p],
    "bts_pack_weight": [_p, _i, _i, _i, _i, _p, _i, _i, _i, _p, _i, _p, _p],
    "bts_unpack_wgrad": [_p, _i, _i, _i, _p, _i, _i, _p, _p, _i, _p],
    "bts_pack_weight_batch": [_p, _i, _l, _i, _p],
    "bts_unpack_wgrad_batch": [_p, _i, _l, _p, _p, _p],
    "bts_nchw_to_nhwc": [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "bts_nhwc_to_nchw": [_p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    "bts_bn_stats_workspace_bytes": [_l, _i],
    "bts_bn_stats": [_p, _i, _i, _l, _i, _p, _p, _p, _p],
    "bts_bn_stats_finalize": [_p, _i, _i, _l, _p, _p, _p],
    "bts_bn_prepare": [_p, _p, _i, _l, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p],
    "bts_affine_act": [_p, _i, _i, _p, _i, _i, _l, _i, _p, _p, _i, _p],
    "bts_bn_bwd_reduce": [_p, _i, _p, _i, _i, _l, _i, _p, _p, _p, _p, _i, _p, _p, _p],
    "bts_bn_bwd_apply": [_p, _i, _p, _i, _p, _i, _i, _l, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p],
    "bts_bn_apply": [C.POINTER(BnDesc), _p],
    "bts_bn_bwd_workspace_bytes": [C.POINTER(BnDesc)],
    "bts_bn_bwd": [C.POINTER(BnDesc), _p, _p, _p],
    "bts_bn_bwd_multi_workspace_bytes": [C.POINTER(BnMultiDesc)],
    "bts_bn_bwd_multi": [C.POINTER(BnMultiDesc), _p, _p],
    "bts_act_bwd": [_p, _i, _i, _p, _i, _i, _p, _i, _i, _l, _i, _i, _f, _p, _l, _i, _p],
    "bts_add_to": [_p, _i, _i, _p, _i, _i, _l, _i, _i, _p],
    "bts_adamw_step": [_p, _p, _p, _p, _p, _i, _l, _f, _f, _f, _f, _f, _f, _f, _p, _p],
    "bts_adamw_advance": [_p, _i, _p],
}
_LONG_RET = {"bts_silog_workspace_bytes", "bts_bn_stats_workspace_bytes", "bts_eval_workspace_bytes", "bts_bn_bwd_workspace_bytes",
             "bts_bn_bwd_multi_workspace_bytes"}
_NO_CHECK = _LONG_RET | {"bts_abi_version", "bts_current_device"}

_lib = None


def load():
    """dlopen the library (once) and attach prototypes.  Raises BtsAmdError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BtsAmdError(
            "libbts_amd.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python bts_amd/build.py`).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = C.c_long if name in _LONG_RET else C.c_int
    if lib.bts_abi_version() != 4:
        raise BtsAmdError("libbts_amd.so ABI version mismatch")
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point and raise on a non-zero status."""
    from . import profiler
    if profiler.ACTIVE is not None and name not in _NO_CHECK:
        note = profiler.take() or (name[4:], "hbm", 0.0, None, 0.0)      # undescribed launches are still timed (zero work)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(load(), name)(*args)
        e.record()
        if rc != ERR_UNSUPPORTED:      # nothing was launched: the caller retries with other launches, which record their own work
            profiler.ACTIVE.add(note[0], note[1], note[2], s, e, note[3], note[4])
        if rc != 0:
            raise BtsAmdError("%s failed: %s (%d)" % (name, ERRORS.get(rc, "?"), rc), rc)
        return rc
    rc = getattr(load(), name)(*args)
    if name not in _NO_CHECK and rc != 0:
        raise BtsAmdError("%s failed: %s (%d)" % (name, ERRORS.get(rc, "?"), rc), rc)
    return rc


def stream_ptr():
    """The current PyTorch HIP stream as a raw hipStream_t."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt):
    if dt == torch.float32:
        return BTS_F32
    if dt == torch.bfloat16:
        return BTS_BF16
    raise BtsAmdError("unsupported activation dtype %s" % dt)


def require_gpu(t):
    if not t.is_cuda:
        raise BtsAmdError("bts_amd kernels run on an MI355X (HIP) device only; got a %s tensor. "
                          "There is no CPU fallback in the product path." % t.device)
