"""Host-side description of one decoder convolution for the implicit-GEMM kernels.

A ``ConvLayer`` owns the static geometry of a conv of pytorch/bts.py (kernel size, dilation,
fused nearest-x2 up-sampling, how its input channels split over the concatenated input
tensors) and turns the PyTorch-layout f32 weight ``[Cout, Cin, kh, kw]`` into the packed
operands of ``bts_conv_fwd`` / its data-gradient, and the packed weight-gradient back.

Up-sampling convs (``upconv``, bts.py:69-80: nearest x2 then 3x3, pad 1) are evaluated as four
2x2 sub-pixel phase convolutions on the LOW-resolution input: output pixel (2i+a, 2j+b) only
ever sees input rows {i-1, i} (a = 0) or {i, i+1} (a = 1), with the 3x3 taps that land on the
same low-res pixel pre-summed.  That is 16/36 of the MACs of the literal formulation and never
materialises the up-sampled tensor.
"""
import ctypes as C

import os

import torch

from . import _lib, profiler
from ._lib import ConvDesc, call, dtype_code, stream_ptr
from .ops import pad_to, pix_stride, vec_of

# row/column structure of the sub-pixel phases: phase parity -> [(low-res offset, [3x3 tap indices])]
_PHASE = {0: [(-1, [0]), (0, [1, 2])], 1: [(0, [0, 1]), (1, [2])]}


def _taps_plain(kk, dil):
    if kk == 1:
        return [(0, 0, 0, 0)], [1]
    taps, masks = [], []
    for ky in range(3):
        for kx in range(3):
            taps.append(((ky - 1) * dil, (kx - 1) * dil, 0, 0))
            masks.append(1 << (ky * 3 + kx))
    return taps, masks


def _taps_up():
    """16 (phase-major) taps of the sub-pixel decomposition: (dy, dx, a, b) and 3x3 source masks."""
    taps, masks = [], []
    for a in (0, 1):
        for b in (0, 1):
            for oy, kys in _PHASE[a]:
                for ox, kxs in _PHASE[b]:
                    taps.append((oy, ox, a, b))
                    masks.append(sum(1 << (ky * 3 + kx) for ky in kys for kx in kxs))
    return taps, masks


def _dn(dtype):
    return "f32" if dtype == torch.float32 else "bf16"


def _cdiv(a, b):
    return (a + b - 1) // b


_CUS = []


def _cu_count():
    if not _CUS:
        _CUS.append(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256)
    return _CUS[0]


def _wide_mode():
    """BTS_CONV_WIDE = 0 | 1 | 2: the one run-time switch of the forward dispatch (csrc/conv_igemm.hip::wide_mode): conv_halo_wide
    off (counter-collection runs: rocprofv3 aborts on its 160 KiB of dynamic LDS) / by the fill heuristic [default] / wherever applicable."""
    try:
        return int(os.environ.get("BTS_CONV_WIDE", "1"))
    except ValueError:
        return 1


def _wide_pays(cout, n, hg, wg):
    """Mirrors launch_halo_wide()'s fill heuristic (csrc/conv_halo_wide.hip)."""
    cus = _cu_count()
    mode = _wide_mode()
    if mode == 0:
        return False
    if mode >= 2:
        return True
    bm = 64 if cout <= 64 else (96 if cout % 96 == 0 and cout % 128 != 0 else 128)
    ntiles = _cdiv(wg, 32) * _cdiv(hg, 8) * n
    nco = _cdiv(cout, bm)
    wgs = ntiles * nco
    rounds = _cdiv(wgs, cus)
    fill = (hg * wg * n / (ntiles * 256.0)) * (cout / float(nco * bm)) * (wgs / float(rounds * cus))
    return fill >= 0.60


def _fwd_kernel(dtype, cout, halo, geom=None, kv=8, up=False):
    """Name of the kernel launch_fwd() (csrc/conv_igemm.hip) picks: profiler label = rocprofv3 kernel family."""
    bf = dtype == torch.bfloat16
    if halo and not up and bf and 32 < cout <= 64 and kv > 8 and geom is not None and _wide_pays(cout, *geom):
        return "conv_halo_wide<bf16,64x256>"
    if halo and cout <= 64:
        return "conv_halo<%s>" % _dn(dtype)
    if halo and not up and bf and kv >= 8 and geom is not None and _wide_pays(cout, *geom):
        if cout % 96 == 0 and cout % 128 != 0:
            return "conv_halo_wide<bf16,96x256>"
        return "conv_halo_wide<bf16,128x256>"
    return "conv_igemm_dma<%s,%s>" % (_dn(dtype), "128x128" if cout > 64 else ("64x128" if cout > 32 else "32x256"))


def _wgrad_kernel(dtype, cout, radius1, up, n, hg, wg, cols=None):
    """Mirrors launch_wgrad() (csrc/conv_wgrad.hip) / launch_wgrad_tr() (csrc/conv_wgrad_tr.hip).  cols = taps per phase x padded
    input channels (T * Ktot)."""
    bf = dtype == torch.bfloat16
    big_map = _cdiv(wg, 32) * _cdiv(hg, 8) * n >= 256
    if radius1 and not up and cout == 1:
        return "conv_wgrad_c1<%s>" % _dn(dtype)
    if radius1 and not up and 1 < cout <= 128 and bf and big_map:
        return "conv_wgrad_halo_tr<bf16>"
    if radius1 and up and cout in (32, 64) and bf and big_map:       # r6: launch_wgrad_halo_tr_up (dz pixel stride == cout)
        return "conv_wgrad_halo_tr_up<bf16>"
    if 32 < cout <= 64 and bf:
        return "conv_wgrad_ring<bf16,64x256>"
    if radius1 and up and cout <= 64 and bf and big_map:
        return "conv_wgrad_halo_up<bf16>"
    if cout > 64 and bf:
        ring_tiles = _cdiv(cout, 128) * _cdiv(cols if cols is not None else 256, 256) * (4 if up else 1)
        if ring_tiles <= 2 * _cu_count():
            return "conv_wgrad_ring<bf16,128x256>"
        return "conv_wgrad_tr<bf16,128x128>"
    return "conv_wgrad<%s,%s>" % (_dn(dtype), "128x128" if cout > 64 else ("64x128k2" if cout > 32 else "32x128k4"))


def _res_enabled():
    """BTS_RES = 0 keeps every weight operand in the [rows][taps][K] layout, i.e. conv_igemm_res off (A/B switch; default on)."""
    return os.environ.get("BTS_RES", "1") != "0"


class ConvLayer:
    """Geometry + weight packing of one conv.  seg_channels: logical channels per input tensor."""

    def __init__(self, name, cout, seg_channels, kk, dil=1, up=False):
        self.name, self.cout, self.kk, self.dil, self.up = name, cout, kk, dil, up
        self.seg_channels = list(seg_channels)
        self.cin = sum(seg_channels)
        if up:
            assert kk == 9 and dil == 1
            self.taps, self.masks = _taps_up()
            self.nphase, self.T = 4, 4
        else:
            self.taps, self.masks = _taps_plain(kk, dil)
            self.nphase, self.T = 1, len(self.taps)
        self._cache = {}
        self._stats_rows = {}      # launch shape -> rows of partial statistics the selected kernel writes (0: none)

    # -- per (dtype, device) tables ----------------------------------------------------------
    def tables(self, dtype, device):
        key = (dtype, str(device))
        tb = self._cache.get(key)
        if tb is None:
            v = vec_of(dtype)
            pads = [pad_to(c, v) for c in self.seg_channels]
            cmap, kinv, seg_rows = [], [], []
            c0 = 0
            for c, cp in zip(self.seg_channels, pads):
                base = len(cmap)
                cmap += list(range(c0, c0 + c)) + [-1] * (cp - c)
                kinv += list(range(base, base + c))
                seg_rows.append(list(range(c0, c0 + c)) + [-1] * (cp - c))
                c0 += c
            tb = dict(
                v=v, pads=pads, ktot=sum(pads), cout_pad=pad_to(self.cout, v),
                cmap=torch.tensor(cmap, dtype=torch.int32, device=device),
                kinv=torch.tensor(kinv, dtype=torch.int32, device=device),
                seg_rows=[torch.tensor(r, dtype=torch.int32, device=device) for r in seg_rows],
                cmap_out=torch.tensor(list(range(self.cout)) + [-1] * (pad_to(self.cout, v) - self.cout),
                                      dtype=torch.int32, device=device),
                masks=(C.c_uint16 * len(self.masks))(*self.masks),
            )
            self._cache[key] = tb
        return tb

    # -- weight layout -------------------------------------------------------------------------
    def frag_layout(self, dtype, rows, K, dgrad=False):
        """bts_conv_desc_t::w_frag of the operand of a forward (rows = Cout, K = padded input channels) or data-gradient launch
        (rows = padded channels of the input segment, K = padded Cout): 1 = MFMA A-fragment order for conv_igemm_res -- bf16, more
        than 64 output rows, at most four 64-deep chunks of K (the kernel keeps the whole pixel operand of a tile resident), and a
        launch that ALWAYS goes to the implicit GEMM whatever the map size: 1x1, dilated 3x3, sub-pixel up-convolutions.  In this
        decoder: the data gradients of the five dense-ASPP 1x1 layers (K = 256), daspp_3's 1x1 forward, reduc8x8's 128 -> 128 layer.
        Radius-1 3x3 layers are decided per geometry inside the library and keep the row-major operand.  Mirrors launch_fwd()
        (csrc/conv_igemm.hip), which refuses a fragment operand it cannot use."""
        if not (_res_enabled() and dtype == torch.bfloat16 and rows > 64 and (self.up or self.kk == 1 or self.dil > 1)):
            return 0
        nphase, Tp = self._launch_taps(dgrad)
        if Tp > 1 and nphase == 1 and K % 64 == 0:        # launch_fwd() runs those channel-chunk-major; the fragment order is tap-major
            return 0
        return 1 if _cdiv(Tp * K, 64) <= 4 else 0

    def frag_bytes(self, rows, K, dgrad):
        """size of a fragment-order operand (zero-initialised by the caller; the pack kernel writes the real entries)"""
        nphase, Tp = self._launch_taps(dgrad)
        return nphase * 4 * _cdiv(rows, 128) * _cdiv(Tp * K, 64) * 4096

    def _launch_taps(self, dgrad):
        """(phases, taps per phase) of the LAUNCH that consumes the operand: the data gradient of an up-convolution runs its 16 taps
        as one phase (input scale 2), everything else as the layer is described"""
        return (1, self.nphase * self.T) if (dgrad and self.up) else (self.nphase, self.T)

    @staticmethod
    def to_frag(wp, nphase):
        """Row-major packed operand [R][nphase*Tp][K] (bf16) -> fragment order, by plain torch indexing: the host-side statement of
        the layout (tests hold the pack kernel's fragment stores against it)."""
        R, Tt, K = wp.shape
        Tp = Tt // nphase
        RT = 4 * _cdiv(R, 128)
        nch = _cdiv(Tp * K, 64)
        dev = wp.device
        wpad = torch.zeros((RT * 32, nphase, Tp * K + 64), dtype=wp.dtype, device=dev)
        wpad[:R, :, :Tp * K] = wp.reshape(R, nphase, Tp * K)
        c = torch.arange(nch, device=dev)[:, None, None, None]
        sidx = torch.arange(4, device=dev)[None, :, None, None]
        g = torch.arange(2, device=dev)[None, None, :, None]
        e = torch.arange(8, device=dev)[None, None, None, :]
        col = 64 * c + 16 * sidx + 8 * g + e
        col = torch.where(col < Tp * K, col, torch.full_like(col, Tp * K)).expand(nch, 4, 2, 8)     # past the end: a zero column
        src = wpad.reshape(RT, 32, nphase, Tp * K + 64).permute(2, 0, 1, 3)                # [ph][rt][row][col]
        out = src[:, :, :, col]                                                            # [ph][rt][row][c][s][g][e]
        return out.permute(0, 1, 3, 4, 5, 2, 6).contiguous().reshape(-1)                   # [ph][rt][c][s][lane = 32 g + row][e]

    # -- weight packing ------------------------------------------------------------------------
    def pack_fwd(self, weight, dtype):
        tb = self.tables(dtype, weight.device)
        ttot = self.nphase * self.T
        out = torch.empty((self.cout, ttot, tb["ktot"]), dtype=dtype, device=weight.device)
        call("bts_pack_weight", C.c_void_p(weight.data_ptr()), self.cout, self.cin, self.kk, 0,
             C.c_void_p(tb["cmap"].data_ptr()), self.cout, tb["ktot"], ttot, tb["masks"], dtype_code(dtype),
             C.c_void_p(out.data_ptr()), stream_ptr())
        return out

    def pack_dgrad(self, weight, dtype, seg):
        """Data-gradient operand for input segment `seg`: [Cseg_pad][Ttot][Cout_pad]."""
        tb = self.tables(dtype, weight.device)
        ttot = self.nphase * self.T
        rows = tb["seg_rows"][seg]
        out = torch.empty((rows.numel(), ttot, tb["cout_pad"]), dtype=dtype, device=weight.device)
        call("bts_pack_weight", C.c_void_p(weight.data_ptr()), self.cout, self.cin, self.kk, 1,
             C.c_void_p(rows.data_ptr()), rows.numel(), tb["cout_pad"], ttot, tb["masks"], dtype_code(dtype),
             C.c_void_p(out.data_ptr()), stream_ptr())
        return out

    def unpack_wgrad(self, dwp, dtype):
        tb = self.tables(dtype, dwp.device)
        ttot = self.nphase * self.T
        shape = (self.cout, self.cin, 3, 3) if self.kk == 9 else (self.cout, self.cin, 1, 1)
        gw = torch.empty(shape, dtype=torch.float32, device=dwp.device)
        call("bts_unpack_wgrad", C.c_void_p(dwp.data_ptr()), self.cout, self.cin, self.kk, C.c_void_p(tb["kinv"].data_ptr()),
             tb["ktot"], ttot, tb["masks"], C.c_void_p(gw.data_ptr()), 0, stream_ptr())
        return gw

    # -- descriptors -----------------------------------------------------------------------------
    def _desc(self, dtype, segs, N, Hg, Wg):
        d = ConvDesc()
        d.dtype = dtype_code(dtype)
        d.N, d.Hg, d.Wg = N, Hg, Wg
        d.nseg = len(segs)
        for i, s in enumerate(segs):
            d.seg[i].ptr = s.data_ptr()
            d.seg[i].C = s.shape[3]
            d.seg[i].stride = pix_stride(s)
        d.Hx, d.Wx = segs[0].shape[1], segs[0].shape[2]
        return d

    def forward(self, segs, wp, out, act, out_scale=1.0, out_scale_n=None, w_frag=0, stats=None):
        """segs: NHWC input tensors (padded channels); wp = pack_fwd(weight); out: NHWC [N,Ho,Wo,>=Cout]
        or a single-channel f32 map [N,Ho,Wo].  w_frag: layout of wp (frag_layout()).
        stats: a list; when the kernel this launch selects has the statistics epilogue (include/bts_amd.h:
        bts_conv_desc_t::stats_ws), the batch statistics (mean, biased var) of `out` -- what ops.bn_stats(out) would compute with
        one more pass over it -- are appended; left empty otherwise (the caller then runs ops.bn_stats when it needs them)."""
        dtype = segs[0].dtype
        N, Hx, Wx, _ = segs[0].shape
        d = self._desc(dtype, segs, N, Hx, Wx)
        d.isc = 1
        d.nphase, d.T = self.nphase, self.T
        for i, (dy, dx, a, b) in enumerate(self.taps):
            d.dy[i], d.dx[i], d.ioy[i], d.iox[i] = dy, dx, 0, 0
        d.w = wp.data_ptr()
        d.w_frag = w_frag
        d.Cout = self.cout
        self._set_out(d, out)
        d.osc = 2 if self.up else 1
        d.act = act
        d.out_scale = out_scale
        d.out_scale_n = out_scale_n.data_ptr() if out_scale_n is not None else None
        d.accumulate = 0
        ws = None
        if stats is not None and out.dim() == 4 and out.dtype == torch.bfloat16:
            key = (dtype, N, Hx, Wx, pix_stride(out), act, w_frag, out_scale, out_scale_n is None)
            rows = self._stats_rows.get(key)
            if rows is None:          # a property of the kernel the descriptor selects: asked once per launch shape
                r = C.c_int(0)
                rows = r.value if _lib.load().bts_conv_fwd_stats_rows(C.byref(d), C.byref(r)) == 0 else 0
                self._stats_rows[key] = rows
            if rows:
                ws = torch.empty((rows, 2, self.cout), dtype=torch.float32, device=out.device)
                d.stats_ws = ws.data_ptr()
        if profiler.ACTIVE is not None:
            M = N * Hx * Wx
            kv = sum(pad_to(c, vec_of(dtype)) for c in self.seg_channels) // vec_of(dtype)
            nb = (sum(t.shape[3] for t in segs) * M * segs[0].element_size() + out.numel() * out.element_size()
                  + wp.numel() * wp.element_size())            # inputs once + output once + packed weights
            profiler.note("conv_igemm_res<bf16,128xN>" if w_frag else _fwd_kernel(dtype, self.cout, self.kk == 9 and self.dil == 1, (N, Hx, Wx), kv, self.up), "mfma",
                          2.0 * M * self.nphase * self.T * self.cin * self.cout, self.name + ".fwd", nb)
        call("bts_conv_fwd", C.byref(d), stream_ptr())
        if ws is not None:
            Mo = out.shape[0] * out.shape[1] * out.shape[2]
            mean = torch.empty(self.cout, dtype=torch.float32, device=out.device)
            var = torch.empty(self.cout, dtype=torch.float32, device=out.device)
            if profiler.ACTIVE is not None:
                profiler.note("bn_stats_finalize", "hbm", ws.numel() * 4)
            call("bts_bn_stats_finalize", ws.data_ptr(), ws.shape[0], self.cout, Mo, mean.data_ptr(), var.data_ptr(), stream_ptr())
            stats.append((mean, var))
        return out

    @staticmethod
    def _set_out(d, out):
        d.y = out.data_ptr()
        d.y_dtype = dtype_code(out.dtype)
        if out.dim() == 3:
            d.y_stride, d.Hy, d.Wy = 1, out.shape[1], out.shape[2]
        else:
            d.y_stride, d.Hy, d.Wy = pix_stride(out), out.shape[1], out.shape[2]

    def dgrad(self, dz, wd, seg_index, gx, accumulate, fold_elu_y=None, w_frag=0):
        """gx (+)= data-gradient w.r.t. input segment seg_index.  dz: NHWC [N,Ho,Wo,Cout_pad].
        fold_elu_y: that segment's forward tensor when it is an ELU output whose gradient this launch completes: the stored
        value is (gx [+ old]) * ELU'(fold_elu_y) (include/bts_amd.h: bts_conv_desc_t::fold_elu_y)."""
        dtype = dz.dtype
        N = gx.shape[0]
        Hg, Wg = gx.shape[1], gx.shape[2]
        d = self._desc(dtype, [dz], N, Hg, Wg)
        if self.up:
            d.isc, d.nphase, d.T = 2, 1, 16
            for i, (dy, dx, a, b) in enumerate(self.taps):
                d.dy[i], d.dx[i], d.ioy[i], d.iox[i] = -dy, -dx, a, b
        else:
            d.isc, d.nphase, d.T = 1, 1, self.T
            for i, (dy, dx, a, b) in enumerate(self.taps):
                d.dy[i], d.dx[i], d.ioy[i], d.iox[i] = -dy, -dx, 0, 0
        d.w = wd.data_ptr()
        d.w_frag = w_frag
        d.Cout = gx.shape[3]
        self._set_out(d, gx)
        d.osc = 1
        d.act = _lib.ACT_NONE
        d.out_scale = 1.0
        d.out_scale_n = None
        d.accumulate = int(accumulate)
        if fold_elu_y is not None:
            if fold_elu_y.shape != gx.shape or fold_elu_y.dtype != gx.dtype:
                raise _lib.BtsAmdError("dgrad: fold_elu_y must have the gradient's shape and dtype")
            d.fold_elu_y, d.fold_elu_stride = fold_elu_y.data_ptr(), pix_stride(fold_elu_y)
        if profiler.ACTIVE is not None:
            cseg = self.seg_channels[seg_index]
            profiler.note("conv_igemm_res<bf16,128xN>" if w_frag else
                          _fwd_kernel(dtype, gx.shape[3], self.kk == 9 and self.dil == 1 and not self.up, (N, Hg, Wg),
                                      pad_to(self.cout, vec_of(dtype)) // vec_of(dtype)), "mfma",
                          2.0 * N * Hg * Wg * len(self.taps) * cseg * self.cout, "%s.dgrad%d" % (self.name, seg_index),
                          dz.numel() * dz.element_size() + wd.numel() * wd.element_size()
                          + gx.numel() * gx.element_size() * (1 + int(bool(accumulate)) + int(fold_elu_y is not None)))
        call("bts_conv_fwd", C.byref(d), stream_ptr())
        return gx

    def dual_dgrad_ok(self, dtype, i, j):
        """Can the data gradients w.r.t. input segments i and j leave as ONE launch (bts_conv_desc_t::y2: second output formed from
        the same dz fragments)?  Domain of conv_halo's register-weight form: bf16, radius-1 3x3, dz of <= 32 channels, segment i of
        exactly 32 channels, segment j of <= 32 (padded) -- conv1 at bts_size 512: 32 -> 32 towards upconv1 and 32 -> 4 towards the
        four depth maps."""
        v = vec_of(dtype)
        return (dtype == torch.bfloat16 and self.kk == 9 and self.dil == 1 and not self.up and pad_to(self.cout, v) <= 32
                and self.seg_channels[i] == 32 and pad_to(self.seg_channels[j], v) <= 32)     # (main output: one full 32-channel block)

    def dgrad_dual(self, dz, wd_i, i, gx_i, acc_i, fold_i, wd_j, j, gx_j, acc_j):
        """gx_i (+)= dgrad w.r.t. segment i (optionally through fold_i's ELU), gx_j (+)= dgrad w.r.t. segment j: one pass over dz."""
        dtype = dz.dtype
        N, Hg, Wg = gx_i.shape[0], gx_i.shape[1], gx_i.shape[2]
        d = self._desc(dtype, [dz], N, Hg, Wg)
        d.isc, d.nphase, d.T = 1, 1, self.T
        for t, (dy, dx, a, b) in enumerate(self.taps):
            d.dy[t], d.dx[t], d.ioy[t], d.iox[t] = -dy, -dx, 0, 0
        d.w = wd_i.data_ptr()
        d.Cout = gx_i.shape[3]
        self._set_out(d, gx_i)
        d.osc = 1
        d.act = _lib.ACT_NONE
        d.out_scale = 1.0
        d.out_scale_n = None
        d.accumulate = int(acc_i)
        if fold_i is not None:
            if fold_i.shape != gx_i.shape or fold_i.dtype != gx_i.dtype:
                raise _lib.BtsAmdError("dgrad_dual: fold_elu_y must have the gradient's shape and dtype")
            d.fold_elu_y, d.fold_elu_stride = fold_i.data_ptr(), pix_stride(fold_i)
        if gx_j.shape[:3] != gx_i.shape[:3] or gx_j.dtype != gx_i.dtype:
            raise _lib.BtsAmdError("dgrad_dual: the two gradients must share pixels and dtype")
        d.w2, d.y2, d.Cout2, d.y2_stride, d.accumulate2 = wd_j.data_ptr(), gx_j.data_ptr(), gx_j.shape[3], pix_stride(gx_j), int(acc_j)
        if profiler.ACTIVE is not None:
            profiler.note("conv_halo<%s>" % _dn(dtype), "mfma",
                          2.0 * N * Hg * Wg * len(self.taps) * (self.seg_channels[i] + self.seg_channels[j]) * self.cout,
                          "%s.dgrad%d+%d" % (self.name, i, j),
                          dz.numel() * dz.element_size() + gx_i.numel() * gx_i.element_size() * (1 + int(bool(acc_i)) + int(fold_i is not None))
                          + gx_j.numel() * gx_j.element_size() * (1 + int(bool(acc_j))))
        call("bts_conv_fwd", C.byref(d), stream_ptr())

    def _wgrad_desc(self, segs, dz):
        dtype = segs[0].dtype
        N, Hx, Wx, _ = segs[0].shape
        d = self._desc(dtype, segs, N, Hx, Wx)
        d.isc = 1
        d.nphase, d.T = self.nphase, self.T
        for i, (dy, dx, a, b) in enumerate(self.taps):
            d.dy[i], d.dx[i], d.ioy[i], d.iox[i] = dy, dx, 0, 0
        d.Cout = self.cout
        d.Hy, d.Wy = dz.shape[1], dz.shape[2]
        d.osc = 2 if self.up else 1
        return d

    def wgrad_groupable(self, dtype, n, hx, wx):
        """Does this layer's weight gradient run on the 128 x 256 ring kernel (csrc/conv_wgrad_tr.hip), i.e. can it join a grouped
        launch (bts_conv_wgrad_group)?  Mirrors launch_wgrad(): the LDS-halo-tile form takes the radius-1 3x3 layers up to 128
        output channels on large maps first."""
        cols = self.T * sum(pad_to(c, vec_of(dtype)) for c in self.seg_channels)
        # only the SMALL layers gain: <= 16 tiles of 128 x 256 (the dense-ASPP layers: 2-9 tiles, split 28-42 ways when launched alone).
        # conv4 / conv5 / daspp_conv (32-128 tiles) fill the chip with a 2-8-way split on their own at ~800 TF; grouped with small
        # layers they measured 664 TF (gpurun r04j: 171 tiles leave the one-round split nothing to balance)
        return (dtype == torch.bfloat16 and not self.up and _cdiv(self.cout, 128) * _cdiv(cols, 256) <= 16
                and _wgrad_kernel(dtype, self.cout, self.kk == 9 and self.dil == 1, self.up, n, hx, wx, cols) == "conv_wgrad_ring<bf16,128x256>")

    def wgrad_packed(self, segs, dz, dwp):
        """Accumulates the packed f32 weight gradient [Cout, nphase*T, Ktot] into `dwp` (caller-zeroed)."""
        dtype = segs[0].dtype
        N, Hx, Wx, _ = segs[0].shape
        d = self._wgrad_desc(segs, dz)
        if profiler.ACTIVE is not None:
            cols = self.T * sum(pad_to(c, vec_of(dtype)) for c in self.seg_channels)
            profiler.note(_wgrad_kernel(dtype, self.cout, self.kk == 9 and self.dil == 1, self.up, N, Hx, Wx, cols), "mfma",
                          2.0 * N * Hx * Wx * self.nphase * self.T * self.cin * self.cout, self.name + ".wgrad",
                          sum(t.numel() for t in segs) * segs[0].element_size() + dz.numel() * dz.element_size() + dwp.numel() * 4 * 2)
        call("bts_conv_wgrad", C.byref(d), C.c_void_p(dz.data_ptr()), pix_stride(dz), C.c_void_p(dwp.data_ptr()), stream_ptr())
        return dwp

    @staticmethod
    def wgrad_group(items):
        """items: [(layer, segs, dz, dwp)] (<= 5, each wgrad_groupable): all their weight gradients in ONE launch."""
        n = len(items)
        descs = [L._wgrad_desc(segs, dz) for L, segs, dz, dwp in items]
        dp = (C.c_void_p * n)(*[C.addressof(d) for d in descs])
        dzp = (C.c_void_p * n)(*[dz.data_ptr() for _, _, dz, _ in items])
        dzs = (C.c_int * n)(*[pix_stride(dz) for _, _, dz, _ in items])
        dwp_ = (C.c_void_p * n)(*[dwp.data_ptr() for _, _, _, dwp in items])
        if profiler.ACTIVE is not None:
            fl = nb = 0.0
            for L, segs, dz, dwp in items:
                N, Hx, Wx, _ = segs[0].shape
                fl += 2.0 * N * Hx * Wx * L.nphase * L.T * L.cin * L.cout
                nb += sum(t.numel() for t in segs) * segs[0].element_size() + dz.numel() * dz.element_size() + dwp.numel() * 4 * 2
            profiler.note("conv_wgrad_ring<bf16,128x256>", "mfma", fl, "+".join(L.name for L, _, _, _ in items) + ".wgrad", nb)
        call("bts_conv_wgrad_group", dp, dzp, dzs, dwp_, n, stream_ptr())

    def wgrad(self, segs, dz):
        """Returns the f32 weight gradient in PyTorch layout."""
        tb = self.tables(segs[0].dtype, dz.device)
        dwp = torch.zeros((self.cout, self.nphase * self.T, tb["ktot"]), dtype=torch.float32, device=dz.device)
        self.wgrad_packed(segs, dz, dwp)
        return self.unpack_wgrad(dwp, segs[0].dtype)
