"""Host side of the fused inference LPG head (csrc/lpg_chain.hip): packs the 1x1-conv weights of a
``reduction_1x1`` chain (pytorch/bts.py:83-108) into MFMA A-fragment order and launches
``bts_lpg_chain_fwd``.

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
             (32, 1, 8), (32, 0, 4), (16, 0, 2)}


def supported(c0, same_first, k):
    return (c0, int(same_first), k) in SUPPORTED


def _pack_layer(w, dtype, first):
    """w: [Cout, Cin] f32 (CPU or GPU) -> flat fragment buffer tensor (same device)."""
    cout, cin = w.shape
    dev = w.device
    tmo = (cout + 31) // 32
    lane = torch.arange(64, device=dev)
    row, g = lane & 31, lane >> 5
    wpad = torch.zeros((tmo * 32, max(cin, 16) + 16), dtype=torch.float32, device=dev)
    wpad[:cout, :cin] = w
    if dtype == torch.bfloat16:
        ks = max(cin, 16) // 16
        e = torch.arange(8, device=dev)
        inner = (8 * g[:, None] + e[None, :]) if first else (torch.tensor(_PERM, device=dev)[None, :] + 4 * g[:, None])   # [64, 8]
        k = 16 * torch.arange(ks, device=dev)[:, None, None] + inner[None]                                               # [ks, 64, 8]
        rows = 32 * torch.arange(tmo, device=dev)[:, None, None, None] + row[None, None, :, None]                        # [tmo,1,64,1]
        frag = wpad[rows.expand(tmo, ks, 64, 8), k[None].expand(tmo, ks, 64, 8)]
        return frag.to(torch.bfloat16).contiguous().view(torch.uint8).flatten()
    ku = max(cin, 8) // 8
    j = torch.arange(4, device=dev)
    k = 8 * torch.arange(ku, device=dev)[:, None, None] + (4 * g[:, None] + j[None, :])[None]                            # [ku, 64, 4]
    rows = 32 * torch.arange(tmo, device=dev)[:, None, None, None] + row[None, None, :, None]
    frag = wpad[rows.expand(tmo, ku, 64, 4), k[None].expand(tmo, ku, 64, 4)]
    return frag.contiguous().view(torch.uint8).flatten()


def pack_chain(weights, dtype):
    """weights: list of Conv2d 1x1 weights [Cout, Cin, 1, 1] in chain order -> uint8 device tensor."""
    parts = [_pack_layer(w.detach().float().reshape(w.shape[0], w.shape[1]), dtype, i == 0) for i, w in enumerate(weights)]
    return torch.cat(parts).contiguous()


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
