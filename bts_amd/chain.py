"""Host side of the fused LPG head kernels (csrc/lpg_chain.hip): packs the 1x1-conv weights of a
``reduction_1x1`` chain (pytorch/bts.py:83-108) into MFMA A-fragment order (and W^T fragments for the recompute
backward) and launches ``bts_lpg_chain_fwd`` / ``bts_lpg_chain_bwd``.

Fragment order: for layer l, output-row tile tm (32 rows) and K step s, one 1 KiB block = 64 lanes x 16 B where
lane = (row & 31) + 32 * g.  bf16: 8 K values per lane; layer 0 uses the natural order k = 16 s + 8 g + e (its
B operand comes from memory), later layers the accumulator order k = 16 s + {0,1,2,3,8,9,10,11}[e] + 4 g (their
B operand is the previous layer's MFMA accumulator, see the kernel header).  f32: 4 K values per lane,
k = 8 u + 4 g + j for every layer.
"""
import ctypes as C

import torch

from . import profiler
from ._lib import call, dtype_code, stream_ptr
from .ops import pix_stride

_PERM = [0, 1, 2, 3, 8, 9, 10, 11]
SUPPORTED = {(128, 1, 8), (128, 0, 4), (64, 0, 2), (32, 0, 1), (64, 1, 8), (64, 0, 4), (32, 0, 2), (16, 0, 1),
             (32, 1, 8), (32, 0, 4), (16, 0, 2), (64, 0, 8), (32, 0, 8), (16, 0, 8), (128, 0, 8)}


def supported(c0, same_first, k):
    return (c0, int(same_first), k) in SUPPORTED


def _frag_coords(cout, cin, dtype, first, dev):
    """(row, k) of every fragment element of a [cout, cin] matrix, in buffer order."""
    tmo = (cout + 31) // 32
    lane = torch.arange(64, device=dev)
    row, g = lane & 31, lane >> 5
    if dtype == torch.bfloat16:
        ks, ne = max(cin, 16) // 16, 8
        e = torch.arange(8, device=dev)
        inner = (8 * g[:, None] + e[None, :]) if first else (torch.tensor(_PERM, device=dev)[None, :] + 4 * g[:, None])   # [64, 8]
        k = 16 * torch.arange(ks, device=dev)[:, None, None] + inner[None]                                               # [ks, 64, 8]
    else:
        ks, ne = max(cin, 8) // 8, 4
        j = torch.arange(4, device=dev)
        k = 8 * torch.arange(ks, device=dev)[:, None, None] + (4 * g[:, None] + j[None, :])[None]                        # [ku, 64, 4]
    rows = 32 * torch.arange(tmo, device=dev)[:, None, None, None] + row[None, None, :, None]                            # [tmo,1,64,1]
    return rows.expand(tmo, ks, 64, ne), k[None].expand(tmo, ks, 64, ne)


def _pack_layer(w, dtype, first):
    """w: [Cout, Cin] f32 (CPU or GPU) -> flat fragment buffer tensor (same device)."""
    cout, cin = w.shape
    rows, k = _frag_coords(cout, cin, dtype, first, w.device)
    wpad = torch.zeros(((cout + 31) // 32 * 32, max(cin, 16) + 16), dtype=torch.float32, device=w.device)
    wpad[:cout, :cin] = w
    frag = wpad[rows, k]
    if dtype == torch.bfloat16:
        frag = frag.to(torch.bfloat16)
    return frag.contiguous().view(torch.uint8).flatten()


class ChainPacker:
    """Per-step repacking of a chain's weights in two gathers: precomputed element indices into the concatenation of
    the flat weights (+ one trailing zero for padding) for the forward fragments and for the W^T fragments."""

    def __init__(self, shapes, dtype, dev):
        self.dtype = dtype
        offs, total = [], 0
        for co, ci in shapes:
            offs.append(total)
            total += co * ci
        self.zero_at = total

        def index(transposed):
            parts = []
            for i, ((co, ci), off) in enumerate(zip(shapes, offs)):
                if transposed:          # matrix W^T: rows = ci, K = co
                    rows, k = _frag_coords(ci, co, dtype, False, dev)
                    src = k * ci + rows
                    ok = (rows < ci) & (k < co)
                else:
                    rows, k = _frag_coords(co, ci, dtype, i == 0, dev)
                    src = rows * ci + k
                    ok = (rows < co) & (k < ci)
                parts.append(torch.where(ok, src + off, torch.full_like(src, total)).flatten())
            return torch.cat(parts).contiguous()
        self.idx_fwd, self.idx_t = index(False), index(True)

    def pack(self, weights, transposed_too):
        flat = torch.cat([w.detach().reshape(-1).float() for w in weights] + [torch.zeros(1, device=weights[0].device)])
        f = flat[self.idx_fwd].to(self.dtype).view(torch.uint8)
        t = flat[self.idx_t].to(self.dtype).view(torch.uint8) if transposed_too else None
        return f, t


def pack_chain(weights, dtype):
    """weights: list of Conv2d 1x1 weights [Cout, Cin, 1, 1] in chain order -> uint8 device tensor."""
    parts = [_pack_layer(w.detach().float().reshape(w.shape[0], w.shape[1]), dtype, i == 0) for i, w in enumerate(weights)]
    return torch.cat(parts).contiguous()


BWD_SUPPORTED = {(64, 2), (32, 1), (64, 4), (32, 2), (16, 1), (32, 4), (16, 2), (64, 8), (32, 8), (16, 8), (128, 4), (128, 8)}


def bwd_supported(c0, same_first, k, dtype):
    """Shapes with a fused recompute backward (csrc/lpg_chain.hip::lpg_chain_bwd_kernel, bf16; lpg_chain_bwd_f32_kernel, f32: the
    parity configuration trains through the same fused kernel design the bf16 headline times)."""
    return dtype in (torch.bfloat16, torch.float32) and not same_first and (c0, k) in BWD_SUPPORTED


def pack_chain_t(weights, dtype):
    """Per layer W^T (rows = input channels, K = output channels in accumulator order) for dA = W^T dz."""
    parts = [_pack_layer(w.detach().float().reshape(w.shape[0], w.shape[1]).t().contiguous(), dtype, False) for w in weights]
    return torch.cat(parts).contiguous()


def chain_bwd(x, frags, frags_t, c0, k, max_depth, grad_out, grad_x, accumulate, grad_w, x_is_elu_output=False):
    """Fused backward: grad_out f32 [N,h*k,w*k] (or [N,h,w] for k = 1) -> grad_x (NHWC like x, written or accumulated)
    and the packed f32 weight gradients `grad_w` (list of [Cout_l, ld_l] views, accumulated)."""
    N, h, w, _ = x.shape
    n = len(grad_w)
    ptrs = (C.c_void_p * n)(*[g.data_ptr() for g in grad_w])
    lds = (C.c_int * n)(*[g.stride(0) for g in grad_w])
    if profiler.ACTIVE is not None:      # x read, grad_out read, grad_x written (+ read when accumulating)
        profiler.note("lpg_head_chain_bwd<k=%d>" % k, "hbm",
                      N * h * w * (c0 * x.element_size() * (3 if accumulate else 2) + 4 * k * k))
    call("bts_lpg_chain_bwd", C.c_void_p(x.data_ptr()), dtype_code(x.dtype), pix_stride(x), c0, C.c_void_p(frags.data_ptr()),
         frags.numel(), C.c_void_p(frags_t.data_ptr()), frags_t.numel(), C.c_void_p(grad_out.data_ptr()),
         C.c_void_p(grad_x.data_ptr()), pix_stride(grad_x), int(bool(accumulate)), int(bool(x_is_elu_output)), ptrs, lds, n,
         N * h * w, h, w, k,
         float(max_depth), stream_ptr())


def chain_fwd(x, frags, c0, same_first, k, max_depth):
    """x: NHWC [N,h,w,C0(+pad)] -> depth [N,h*k,w*k] f32 (k in 8/4/2, already / max_depth) or sigmoid map [N,h,w] (k = 1)."""
    N, h, w, _ = x.shape
    out = torch.empty((N, h * k, w * k) if k > 1 else (N, h, w), dtype=torch.float32, device=x.device)
    if profiler.ACTIVE is not None:      # algorithmic bytes: C0 input channels read once + k*k f32 written
        profiler.note("lpg_head_chain_fwd<k=%d>" % k, "hbm", N * h * w * (c0 * x.element_size() + 4 * k * k))
    call("bts_lpg_chain_fwd", C.c_void_p(x.data_ptr()), dtype_code(x.dtype), pix_stride(x), c0, int(same_first),
         C.c_void_p(frags.data_ptr()), frags.numel(), C.c_void_p(out.data_ptr()), N * h * w, h, w, k, float(max_depth),
         stream_ptr())
    return out
