"""Tensor-level wrappers over the libbts_amd.so C ABI (no autograd here).

Activations are NHWC torch tensors ``[N, H, W, C]`` (f32 or bf16) whose last dim is contiguous
and whose pixels are evenly strided, so a channel slice ``t[..., a:b]`` of a wider buffer is a
valid argument (pointer + pixel stride).  Every function enqueues on the current PyTorch HIP
stream and raises ``BtsAmdError`` on any failure -- there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib, profiler
from ._lib import ACT_ELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BtsAmdError, call, dtype_code, stream_ptr  # noqa: F401


def vec_of(dtype):
    return 4 if dtype == torch.float32 else 8


def pad_to(c, v):
    return (c + v - 1) // v * v


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def pix_stride(t):
    """Pixel stride (elements) of an NHWC tensor/slice; validates the layout."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise BtsAmdError("expected NHWC tensor with contiguous channels, got strides %s" % (t.stride(),))
    s = t.stride(2)
    N, H, W, _ = t.shape
    if (H > 1 and t.stride(1) != W * s) or (N > 1 and t.stride(0) != H * W * s):
        raise BtsAmdError("NHWC tensor pixels are not evenly strided: %s %s" % (t.shape, t.stride()))
    return s


def npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


# ---------------------------------------------------------------------------------------------
# LPG op boundary (TF-op layout): plane_eq [B,h,w,4] f32 -> depth [B,h*k,w*k] f32
# ---------------------------------------------------------------------------------------------
def lpg_fwd(plane_eq, k, depth_div=1.0, focal=None, out=None):
    _lib.require_gpu(plane_eq)
    B, h, w, four = plane_eq.shape
    assert four == 4 and plane_eq.dtype == torch.float32 and plane_eq.is_contiguous()
    if out is None:
        out = torch.empty((B, h * k, w * k), dtype=torch.float32, device=plane_eq.device)
    call("bts_lpg_fwd", _p(plane_eq), _p(focal), _p(out), B, h, w, k, float(depth_div), stream_ptr())
    return out


def lpg_bwd(grad_depth, plane_eq, k, depth_div=1.0, focal=None, out=None):
    B, h, w, _ = plane_eq.shape
    grad_depth = grad_depth.contiguous()
    g = torch.empty_like(plane_eq) if out is None else out
    call("bts_lpg_bwd", _p(grad_depth), _p(plane_eq), _p(focal), _p(g), B, h, w, k, float(depth_div), stream_ptr())
    return g


def _multi_args(eqs, ks, divs):
    n = len(eqs)
    B = (C.c_int * n)(*[e.shape[0] for e in eqs])
    h = (C.c_int * n)(*[e.shape[1] for e in eqs])
    w = (C.c_int * n)(*[e.shape[2] for e in eqs])
    kk = (C.c_int * n)(*ks)
    dv = (C.c_float * n)(*[float(d) for d in divs])
    return n, B, h, w, kk, dv


def _check_multi(plane_eqs, ks, full_maps, coarse_outs, what):
    """Shared argument checks of the multi-problem LPG launches: the kernels take raw pointers, so a wrong dtype / stride / shape
    would be silently wrong results, not an error."""
    n = len(plane_eqs)
    if not (1 <= n <= 4 and len(ks) == n):
        raise BtsAmdError("%s: 1..4 problems with one k each (got %d plane_eqs, %d ks)" % (what, n, len(ks)))
    for e, k in zip(plane_eqs, ks):
        _lib.require_gpu(e)
        if not (e.dim() == 4 and e.shape[3] == 4 and e.dtype == torch.float32 and e.is_contiguous() and int(k) >= 1):
            raise BtsAmdError("%s: plane_eq must be contiguous f32 [B,h,w,4] (got %s %s)" % (what, tuple(e.shape), e.dtype))
    for name, ts, shape_of in (("depth map", full_maps, lambda e, k: (e.shape[0], e.shape[1] * k, e.shape[2] * k)),
                               ("plane gradient", coarse_outs, lambda e, k: tuple(e.shape))):
        if ts is None:
            continue
        if len(ts) != n:
            raise BtsAmdError("%s: %d %ss for %d problems" % (what, len(ts), name, n))
        for t, e, k in zip(ts, plane_eqs, ks):
            if not (tuple(t.shape) == shape_of(e, k) and t.dtype == torch.float32 and t.is_contiguous() and t.device == e.device):
                raise BtsAmdError("%s: %s must be contiguous f32 %s on %s (got %s %s)" % (what, name, shape_of(e, k), e.device,
                                                                                        tuple(t.shape), t.dtype))


def lpg_fwd_multi(plane_eqs, ks, depth_divs=None, outs=None):
    """Several LPG problems (<= 4; e.g. the k = 8 / 4 / 2 heads of one batch) in ONE launch (bts_lpg_fwd_multi)."""
    _check_multi(plane_eqs, ks, outs, None, "lpg_fwd_multi")
    divs = depth_divs or [1.0] * len(ks)
    if outs is None:
        outs = [torch.empty((e.shape[0], e.shape[1] * k, e.shape[2] * k), dtype=torch.float32, device=e.device) for e, k in zip(plane_eqs, ks)]
    n, B, h, w, kk, dv = _multi_args(plane_eqs, ks, divs)
    call("bts_lpg_fwd_multi", n, (C.c_void_p * n)(*[e.data_ptr() for e in plane_eqs]), (C.c_void_p * n)(*[o.data_ptr() for o in outs]),
         B, h, w, kk, dv, stream_ptr())
    return outs


def lpg_bwd_multi(grad_depths, plane_eqs, ks, depth_divs=None, outs=None):
    divs = depth_divs or [1.0] * len(ks)
    grad_depths = [g.contiguous() for g in grad_depths]
    _check_multi(plane_eqs, ks, grad_depths, outs, "lpg_bwd_multi")
    if outs is None:
        outs = [torch.empty_like(e) for e in plane_eqs]
    n, B, h, w, kk, dv = _multi_args(plane_eqs, ks, divs)
    call("bts_lpg_bwd_multi", n, (C.c_void_p * n)(*[g.data_ptr() for g in grad_depths]), (C.c_void_p * n)(*[e.data_ptr() for e in plane_eqs]),
         (C.c_void_p * n)(*[o.data_ptr() for o in outs]), B, h, w, kk, dv, stream_ptr())
    return outs


def lpg_head_fwd(raw, k, max_depth, want_plane=False):
    """raw [B,h,w,>=3] f32 (channel-contiguous) -> depth [B,h*k,w*k] f32 (already / max_depth)."""
    _lib.require_gpu(raw)
    B, h, w, _ = raw.shape
    assert raw.dtype == torch.float32
    depth = torch.empty((B, h * k, w * k), dtype=torch.float32, device=raw.device)
    plane = torch.empty((B, h, w, 4), dtype=torch.float32, device=raw.device) if want_plane else None
    if profiler.ACTIVE is not None:   # algorithmic bytes: read 3 f32 per cell (stride-4 records: 16 B), write k*k f32
        profiler.note("lpg_head_fwd<k=%d>" % k, "hbm", B * h * w * (16 + 4 * k * k))
    call("bts_lpg_head_fwd", _p(raw), pix_stride(raw), _p(depth), _p(plane), B, h, w, k, float(max_depth), stream_ptr())
    return (depth, plane) if want_plane else depth


def lpg_head_bwd(raw, grad_depth, k, max_depth, grad_dtype, grad_pad):
    B, h, w, _ = raw.shape
    g = torch.empty((B, h, w, grad_pad), dtype=grad_dtype, device=raw.device)
    if profiler.ACTIVE is not None:
        profiler.note("lpg_head_bwd<k=%d>" % k, "hbm", B * h * w * (16 + 4 * k * k + grad_pad * g.element_size()))
    call("bts_lpg_head_bwd", _p(raw), pix_stride(raw), _p(grad_depth), _p(g), dtype_code(grad_dtype), grad_pad, grad_pad,
         B, h, w, k, float(max_depth), stream_ptr())
    return g


def plane_fwd(raw, max_depth):
    """raw [B,h,w,>=3] f32 -> un-normalised plane parameters [B,h,w,4] f32 (bts.py:112-120)."""
    _lib.require_gpu(raw)
    B, h, w, _ = raw.shape
    assert raw.dtype == torch.float32
    plane = torch.empty((B, h, w, 4), dtype=torch.float32, device=raw.device)
    if profiler.ACTIVE is not None:
        profiler.note("plane_fwd", "hbm", B * h * w * 32)
    call("bts_plane_fwd", _p(raw), pix_stride(raw), _p(plane), B * h * w, float(max_depth), stream_ptr())
    return plane


def plane_bwd(raw, grad_plane, max_depth, grad_dtype, grad_pad):
    B, h, w, _ = raw.shape
    g = torch.empty((B, h, w, grad_pad), dtype=grad_dtype, device=raw.device)
    if profiler.ACTIVE is not None:
        profiler.note("plane_bwd", "hbm", B * h * w * (32 + grad_pad * g.element_size()))
    call("bts_plane_bwd", _p(raw), pix_stride(raw), _p(grad_plane.contiguous()), _p(g), dtype_code(grad_dtype), grad_pad, grad_pad,
         B * h * w, float(max_depth), stream_ptr())
    return g


def pack_maps(maps, ds, N, H, W, dtype):
    """maps: list (<=4) of f32 [N, H*ds, W*ds] tensors -> NHWC [N,H,W,VEC] slot buffer."""
    Cp = vec_of(dtype)
    dst = torch.empty((N, H, W, Cp), dtype=dtype, device=maps[0].device)
    n = len(maps)
    src = (C.c_void_p * n)(*[m.data_ptr() for m in maps])
    dsa = (C.c_int * n)(*ds)
    if profiler.ACTIVE is not None:
        profiler.note("pack_maps", "hbm", N * H * W * (4 * n + Cp * dst.element_size()))
    call("bts_pack_maps", src, dsa, n, _p(dst), dtype_code(dtype), Cp, Cp, N, H, W, stream_ptr())
    return dst


def unpack_maps(gdst, gmaps, ds):
    """Adjoint of pack_maps: accumulates channel s of gdst into gmaps[s] (f32, strided by ds[s])."""
    N, H, W, _ = gdst.shape
    n = len(gmaps)
    dst = (C.c_void_p * n)(*[m.data_ptr() for m in gmaps])
    dsa = (C.c_int * n)(*ds)
    if profiler.ACTIVE is not None:
        profiler.note("unpack_maps", "hbm", N * H * W * (8 * n + gdst.shape[3] * gdst.element_size()))
    call("bts_unpack_maps", _p(gdst), dtype_code(gdst.dtype), pix_stride(gdst), dst, dsa, n, N, H, W, stream_ptr())


# ---------------------------------------------------------------------------------------------
# get_depth: 3x3 convolution to one channel + sigmoid * scale (csrc/conv_c1.hip)
# ---------------------------------------------------------------------------------------------
def conv_c1_supported(x, w=None):
    """Domain of the streaming one-output-channel kernels.  w (f32 [1, Cin, 3, 3], PyTorch layout) must have exactly x's channel
    count: the kernels read w[c * 9 + tap] for every channel of x, padding included."""
    ok = x.dim() == 4 and x.shape[3] <= 64 and (x.shape[3] // vec_of(x.dtype)) in (2, 4, 8, 16) \
        and x.shape[3] % vec_of(x.dtype) == 0
    if w is not None:
        ok = ok and w.dim() == 4 and w.shape[0] == 1 and w.shape[1] == x.shape[3] and w.shape[2] == 3 and w.shape[3] == 3
    return ok


def _c1_check(x, w, what):
    if not conv_c1_supported(x, w):
        raise BtsAmdError("%s: x %s / w %s outside the streaming kernels' domain (w must be [1, C, 3, 3] with C = x's channel "
                          "count, padding included)" % (what, tuple(x.shape), tuple(w.shape)))


def conv3x3_c1_fwd(x, w, out_scale, out_scale_n=None):
    """x NHWC [N,H,W,C]; w f32 [1,C,3,3] (PyTorch layout) -> f32 map [N,H,W] = sigmoid(conv3x3) * out_scale * out_scale_n[n]."""
    _lib.require_gpu(x)
    _c1_check(x, w, "conv3x3_c1_fwd")
    N, H, W, Cc = x.shape
    y = torch.empty((N, H, W), dtype=torch.float32, device=x.device)
    if profiler.ACTIVE is not None:
        profiler.note("conv_c1_fwd", "hbm", N * H * W * (Cc * x.element_size() + 4), "get_depth.fwd")
    call("bts_conv3x3_c1_fwd", _p(x), dtype_code(x.dtype), pix_stride(x), Cc, _p(w), _p(y), N, H, W, float(out_scale),
         _p(out_scale_n), stream_ptr())
    return y


def conv3x3_c1_dgrad(grad_y, y, w, gx, accumulate, out_scale, out_scale_n=None, fold_elu_y=None):
    """gx (+)= data gradient of conv3x3_c1_fwd (sigmoid derivative included), optionally through the ELU of fold_elu_y."""
    _c1_check(gx, w, "conv3x3_c1_dgrad")
    N, H, W, Cc = gx.shape
    if profiler.ACTIVE is not None:
        es = gx.element_size()
        profiler.note("conv_c1_dgrad", "hbm", N * H * W * (8 + Cc * es * (1 + int(bool(accumulate)) + int(fold_elu_y is not None))),
                      "get_depth.dgrad")
    call("bts_conv3x3_c1_dgrad", _p(grad_y), _p(y), _p(w), _p(gx), dtype_code(gx.dtype), pix_stride(gx), Cc, int(bool(accumulate)),
         _p(fold_elu_y), pix_stride(fold_elu_y) if fold_elu_y is not None else 0, N, H, W, float(out_scale), _p(out_scale_n),
         stream_ptr())
    return gx


def conv3x3_c1_wgrad(grad_y, y, x, dwp, out_scale, out_scale_n=None):
    """dwp [1, 9, Ktot] f32 (packed weight-gradient layout, caller-zeroed) += weight gradient of conv3x3_c1_fwd; the sigmoid
    derivative is formed in the kernel from (grad_y, y)."""
    N, H, W, Cc = x.shape
    if dwp.dim() != 3 or dwp.shape[0] != 1 or dwp.shape[1] != 9 or dwp.shape[2] < Cc or dwp.dtype != torch.float32:
        raise BtsAmdError("conv3x3_c1_wgrad: dwp must be f32 [1, 9, Ktot >= C], got %s %s" % (tuple(dwp.shape), dwp.dtype))
    if profiler.ACTIVE is not None:
        profiler.note("conv_c1_wgrad", "hbm", N * H * W * (8 + Cc * x.element_size()), "get_depth.wgrad")
    call("bts_conv3x3_c1_wgrad", _p(grad_y), _p(y), _p(x), dtype_code(x.dtype), pix_stride(x), Cc, _p(dwp), int(dwp.shape[2]),
         N, H, W, float(out_scale), _p(out_scale_n), stream_ptr())
    return dwp


# ---------------------------------------------------------------------------------------------
# silog
# ---------------------------------------------------------------------------------------------
def _aligned(t, nbytes):
    """The silog kernels take 16-byte vectors of est / gt / grad and 4-byte words of the mask: an already-contiguous VIEW with an
    odd storage offset (a sliced batch) is copied to a fresh allocation instead of being rejected by the launcher."""
    return t if t is None or t.data_ptr() % nbytes == 0 else t.clone(memory_format=torch.contiguous_format)


def silog_fwd(est, gt, mask, variance_focus, gt_threshold=0.0):
    _lib.require_gpu(est)
    est, gt, mask = _aligned(est, 16), _aligned(gt, 16), _aligned(mask, 4)
    n = est.numel()
    ws = torch.empty(call("bts_silog_workspace_bytes", n) // 8, dtype=torch.float64, device=est.device)
    stats = torch.empty(3, dtype=torch.float64, device=est.device)
    loss = torch.empty(1, dtype=torch.float32, device=est.device)
    if profiler.ACTIVE is not None:       # est + gt (+ 1-byte mask) read once
        profiler.note("silog_fwd", "hbm", n * (8 + (1 if mask is not None else 0)))
    call("bts_silog_fwd", _p(est), _p(gt), _p(mask), float(gt_threshold), n, float(variance_focus), _p(ws), _p(stats),
         _p(loss), stream_ptr())
    return loss, stats


def silog_bwd(est, gt, mask, variance_focus, stats, loss, grad_loss, gt_threshold=0.0):
    est, gt, mask = _aligned(est, 16), _aligned(gt, 16), _aligned(mask, 4)
    g = torch.empty(est.shape, dtype=est.dtype, device=est.device)
    if profiler.ACTIVE is not None:
        profiler.note("silog_bwd", "hbm", est.numel() * (12 + (1 if mask is not None else 0)))
    call("bts_silog_bwd", _p(est), _p(gt), _p(mask), float(gt_threshold), est.numel(), float(variance_focus), _p(stats),
         _p(loss), _p(grad_loss), _p(g), stream_ptr())
    return g


# ---------------------------------------------------------------------------------------------
# layout / elementwise / BN
# ---------------------------------------------------------------------------------------------
def nchw_to_nhwc(src, dtype, relu=False, c_pad=None):
    """src f32/bf16 [N,C,H,W] contiguous -> NHWC [N,H,W,c_pad or C] in `dtype` (pad channels zero)."""
    _lib.require_gpu(src)
    N, Cc, H, W = src.shape
    Cp = c_pad or Cc
    dst = (torch.zeros if Cp != Cc else torch.empty)((N, H, W, Cp), dtype=dtype, device=src.device)
    if profiler.ACTIVE is not None:
        profiler.note("nchw_to_nhwc", "hbm", N * Cc * H * W * (src.element_size() + dst.element_size()))
    call("bts_nchw_to_nhwc", _p(src), dtype_code(src.dtype), _p(dst), dtype_code(dtype), Cp, N, Cc, H, W, int(relu), stream_ptr())
    return dst


def nhwc_to_nchw(src, Cc, relu_src=None, out_dtype=torch.float32):
    N, H, W, _ = src.shape
    dst = torch.empty((N, Cc, H, W), dtype=out_dtype, device=src.device)
    if profiler.ACTIVE is not None:
        profiler.note("nhwc_to_nchw", "hbm", N * Cc * H * W * (src.element_size() + dst.element_size() * (2 if relu_src is not None else 1)))
    call("bts_nhwc_to_nchw", _p(src), dtype_code(src.dtype), pix_stride(src), _p(dst), dtype_code(out_dtype), _p(relu_src),
         N, Cc, H, W, stream_ptr())
    return dst


def bn_stats(x):
    M, Cc = npix(x), x.shape[3]
    ws = torch.empty(call("bts_bn_stats_workspace_bytes", M, Cc) // 4, dtype=torch.float32, device=x.device)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    var = torch.empty(Cc, dtype=torch.float32, device=x.device)
    if profiler.ACTIVE is not None:
        profiler.note("bn_stats", "hbm", M * Cc * x.element_size())
    call("bts_bn_stats", _p(x), dtype_code(x.dtype), pix_stride(x), M, Cc, _p(ws), _p(mean), _p(var), stream_ptr())
    return mean, var


def bn_prepare(mean, var, M, gamma, beta, eps, momentum=0.0, running_mean=None, running_var=None):
    Cc = mean.numel()
    invstd = torch.empty_like(mean)
    scale = torch.empty_like(mean)
    shift = torch.empty_like(mean)
    if profiler.ACTIVE is not None:
        profiler.note("bn_prepare", "hbm", Cc * 4 * 8)
    call("bts_bn_prepare", _p(mean), _p(var), Cc, M, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
         _p(running_var), _p(invstd), _p(scale), _p(shift), stream_ptr())
    return invstd, scale, shift


def affine_act(x, scale, shift, act, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if profiler.ACTIVE is not None:
        profiler.note("affine_act", "hbm", npix(x) * x.shape[3] * (x.element_size() + out.element_size()))
    call("bts_affine_act", _p(x), dtype_code(x.dtype), pix_stride(x), _p(out), dtype_code(out.dtype), pix_stride(out),
         npix(x), x.shape[3], _p(scale), _p(shift), act, stream_ptr())
    return out


def bn_bwd(dy, x, mean, invstd, gamma, beta, relu, dx, accumulate, use_batch_stats=True):
    """Returns (dbeta, dgamma) sums and writes / accumulates dx."""
    M, Cc = npix(x), x.shape[3]
    sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    ws = torch.empty(call("bts_bn_stats_workspace_bytes", M, Cc) // 4, dtype=torch.float32, device=x.device)
    if profiler.ACTIVE is not None:
        profiler.note("bn_bwd_reduce", "hbm", M * Cc * x.element_size() * 2)
    call("bts_bn_bwd_reduce", _p(dy), pix_stride(dy), _p(x), pix_stride(x), dtype_code(x.dtype), M, Cc, _p(mean), _p(invstd),
         _p(gamma), _p(beta), int(relu), _p(ws), _p(sums), stream_ptr())
    if profiler.ACTIVE is not None:
        profiler.note("bn_bwd_apply", "hbm", M * Cc * x.element_size() * (4 if accumulate else 3))
    call("bts_bn_bwd_apply", _p(dy), pix_stride(dy), _p(x), pix_stride(x), _p(dx), pix_stride(dx), dtype_code(x.dtype), M, Cc,
         _p(mean), _p(invstd), _p(gamma), _p(beta), int(relu), _p(sums), int(use_batch_stats), int(accumulate), stream_ptr())
    return sums[0], sums[1]


def _bn_desc(xs, stats, gamma, beta, eps, relu):
    d = _lib.BnDesc()
    d.dtype = dtype_code(xs[0].dtype)
    d.nseg = len(xs)
    d.M = npix(xs[0])
    for i, (x, (mean, var)) in enumerate(zip(xs, stats)):
        d.seg[i].x = x.data_ptr()
        d.seg[i].mean, d.seg[i].var = mean.data_ptr(), var.data_ptr()
        d.seg[i].C, d.seg[i].x_stride = x.shape[3], pix_stride(x)
    d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
    d.eps = float(eps)
    d.relu = int(bool(relu))
    return d


def bn_apply(xs, stats, gamma, beta, eps, relu, out, out2=None, momentum=0.0, running_mean=None, running_var=None, tag=None):
    """BatchNorm(+ReLU) of the channel concatenation of `xs` (NHWC tensors, same pixels) into `out` in ONE launch; `stats` holds
    each tensor's (mean, var).  out2 (optional) receives relu(out).  running_*: train-mode update of the [sum C] buffers."""
    if len(xs) > _lib.BN_MAX_SEG:
        raise BtsAmdError("bn_apply: at most %d concatenated tensors" % _lib.BN_MAX_SEG)
    _lib.require_gpu(xs[0])
    d = _bn_desc(xs, stats, gamma, beta, eps, relu)
    d.momentum = float(momentum)
    d.running_mean = running_mean.data_ptr() if running_mean is not None else None
    d.running_var = running_var.data_ptr() if running_var is not None else None
    d.y, d.y_stride = out.data_ptr(), pix_stride(out)
    if out2 is not None:
        d.y2, d.y2_stride = out2.data_ptr(), pix_stride(out2)
    if profiler.ACTIVE is not None:
        ctot = sum(x.shape[3] for x in xs)
        profiler.note("bn_apply", "hbm", npix(xs[0]) * ctot * xs[0].element_size() * (3 if out2 is not None else 2), tag)
    call("bts_bn_apply", C.byref(d), stream_ptr())
    return out


def bn_bwd_ms(dy, xs, dxs, accs, stats, gamma, beta, eps, relu, use_batch_stats, elu_x=False, tag=None):
    """Backward of bn_apply: dy = gradient of the normalised concatenation; dxs[i] (+)= gradient of xs[i]; returns
    (dbeta, dgamma) over the concatenated channels.  elu_x: see include/bts_amd.h (ELU derivative of the producer folded in)."""
    d = _bn_desc(xs, stats, gamma, beta, eps, relu)
    for i, (dx, acc) in enumerate(zip(dxs, accs)):
        d.seg[i].dx, d.seg[i].dx_stride, d.seg[i].accumulate = dx.data_ptr(), pix_stride(dx), int(bool(acc))
    d.y, d.y_stride = dy.data_ptr(), pix_stride(dy)
    d.use_batch_stats = int(bool(use_batch_stats))
    d.elu_x = int(bool(elu_x))
    ctot = sum(x.shape[3] for x in xs)
    ws = torch.empty(call("bts_bn_bwd_workspace_bytes", C.byref(d)) // 4, dtype=torch.float32, device=dy.device)
    sums = torch.empty((2, ctot), dtype=torch.float32, device=dy.device)
    if profiler.ACTIVE is not None:      # reduction pass: dy + x; apply pass: dy + x read, dx written (+ read when accumulating)
        es = xs[0].element_size()
        profiler.note("bn_bwd", "hbm", npix(xs[0]) * es * sum(x.shape[3] * (6 if a else 5) for x, a in zip(xs, accs)), tag)
    call("bts_bn_bwd", C.byref(d), _p(ws), _p(sums), stream_ptr())
    return sums[0], sums[1]


def bn_bwd_multi(tensors, eps, relu, use_batch_stats, tag=None):
    """Backward of SEVERAL BatchNorm(+ReLU) layers that all normalised the same tensors (the dense-ASPP first_bn layers,
    bts.py:51-66, 211-218) in one reduction + one apply pass: dx (+)= the sum of their input gradients.
    tensors: list (<= 3) of (x, mean, var, dx, accumulate, contribs) over the same pixels, every one with the SAME number (<= 4) of
    contribs = (dy, gamma, beta, dbeta, dgamma): dy = the NHWC channel SLICE [.., c0:c0+C] of that BatchNorm's output gradient,
    gamma / beta / dbeta / dgamma = the same channel range of its f32 parameter / gradient arrays (dbeta, dgamma are written)."""
    nt = len(tensors)
    if not 1 <= nt <= _lib.BN_MULTI_TENSORS:
        raise BtsAmdError("bn_bwd_multi: 1..%d tensors per launch" % _lib.BN_MULTI_TENSORS)
    x0 = tensors[0][0]
    n = len(tensors[0][5])
    if not 1 <= n <= _lib.BN_MAX_MULTI:
        raise BtsAmdError("bn_bwd_multi: 1..%d BatchNorms per launch" % _lib.BN_MAX_MULTI)
    _lib.require_gpu(x0)
    d = _lib.BnMultiDesc()
    d.dtype, d.n, d.nt, d.relu, d.M = dtype_code(x0.dtype), n, nt, int(bool(relu)), npix(x0)
    d.eps, d.use_batch_stats = float(eps), int(bool(use_batch_stats))
    passes = 0
    for ti, (x, mean, var, dx, accumulate, contribs) in enumerate(tensors):
        if len(contribs) != n or x.dtype != x0.dtype or npix(x) != npix(x0) or dx.dtype != x.dtype or tuple(dx.shape) != tuple(x.shape):
            raise BtsAmdError("bn_bwd_multi: tensors of one launch share dtype, pixels and the number of BatchNorms")
        for t in (mean, var):
            if t.dtype != torch.float32 or t.numel() != x.shape[3] or not t.is_contiguous():
                raise BtsAmdError("bn_bwd_multi: statistics must be contiguous f32 [%d]" % x.shape[3])
        T = d.t[ti]
        T.x, T.x_stride, T.C = x.data_ptr(), pix_stride(x), x.shape[3]
        T.mean, T.var = mean.data_ptr(), var.data_ptr()
        T.dx, T.dx_stride, T.accumulate = dx.data_ptr(), pix_stride(dx), int(bool(accumulate))
        for i, (dy, g, b, db, dg) in enumerate(contribs):
            if dy.dtype != x.dtype or tuple(dy.shape) != tuple(x.shape):
                raise BtsAmdError("bn_bwd_multi: gradient slice %s %s does not match x %s %s" % (tuple(dy.shape), dy.dtype, tuple(x.shape), x.dtype))
            for t in (g, b, db, dg):
                if t.dtype != torch.float32 or t.numel() != x.shape[3] or not t.is_contiguous():
                    raise BtsAmdError("bn_bwd_multi: per-channel arrays must be contiguous f32 [%d]" % x.shape[3])
            T.c[i].dy, T.c[i].dy_stride = dy.data_ptr(), pix_stride(dy)
            T.c[i].gamma, T.c[i].beta, T.c[i].dbeta, T.c[i].dgamma = g.data_ptr(), b.data_ptr(), db.data_ptr(), dg.data_ptr()
        passes += x.shape[3] * (2 * (1 + n) + (2 if accumulate else 1))
    ws = torch.empty(call("bts_bn_bwd_multi_workspace_bytes", C.byref(d)) // 4, dtype=torch.float32, device=x0.device)
    if profiler.ACTIVE is not None:      # reduction: x + n dy; apply: x + n dy read, dx written (+ read when accumulating)
        profiler.note("bn_bwd_multi", "hbm", npix(x0) * x0.element_size() * passes, tag)
    call("bts_bn_bwd_multi", C.byref(d), _p(ws), stream_ptr())


def act_bwd(dy, y, act, out=None, out_dtype=None, out_channels=None, y_scale=1.0, y_scale_n=None, accumulate=False):
    """dz (+)= dy * act'(y).  dy/y may be NHWC [N,H,W,C] or single-channel maps [N,H,W]."""
    if dy.dim() == 3:
        N, H, W = dy.shape
        Cc, dys, ys, M = 1, 1, 1, N * H * W
        if out is None:
            oc = out_channels or 1
            out = torch.zeros((N, H, W, oc), dtype=out_dtype or dy.dtype, device=dy.device)
    else:
        Cc, dys, ys, M = dy.shape[3], pix_stride(dy), pix_stride(y), npix(dy)
        if out is None:
            out = torch.empty(dy.shape, dtype=out_dtype or dy.dtype, device=dy.device)
    ppi = M // dy.shape[0]
    if profiler.ACTIVE is not None:
        profiler.note("act_bwd", "hbm", M * Cc * (dy.element_size() + y.element_size() + out.element_size() * (2 if accumulate else 1)))
    call("bts_act_bwd", _p(dy), dtype_code(dy.dtype), dys, _p(y), dtype_code(y.dtype), ys, _p(out), dtype_code(out.dtype),
         pix_stride(out), M, Cc, act, float(y_scale), _p(y_scale_n), ppi, int(bool(accumulate)), stream_ptr())
    return out


def add_to(x, y, accumulate=True):
    if profiler.ACTIVE is not None:
        profiler.note("add_to", "hbm", npix(x) * x.shape[3] * (x.element_size() + y.element_size() * (2 if accumulate else 1)))
    call("bts_add_to", _p(x), dtype_code(x.dtype), pix_stride(x), _p(y), dtype_code(y.dtype), pix_stride(y), npix(x), x.shape[3],
         int(accumulate), stream_ptr())
    return y
