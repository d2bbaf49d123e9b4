"""Explicit forward/backward executor of the BTS decoder on the HIP kernels.

This is the replacement for the body of ``bts.forward`` (pytorch/bts.py:196-266) and for what
PyTorch autograd would record for it.  Instead of a tracing compiler or ~700 eager ATen
launches, the decoder is a short static schedule of C-ABI calls; the forward pass appends
one closure per op to a tape and the backward pass replays the tape in reverse, accumulating
gradients directly into per-tensor gradient buffers (the data-gradient kernels have a
read-add-write epilogue, so fan-out never needs a separate add pass).

Activations are NHWC in ``dtype`` (f32 for parity, bf16 for throughput); parameters, BatchNorm
statistics, LPG heads and the five outputs are f32.  torch.cat never happens: every conv reads
its concatenated input as a list of segments.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import chain as chain_mod
from . import ops, profiler
from ._lib import ACT_ELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BtsAmdError
from .conv import ConvLayer
from .ops import pad_to, vec_of

KITTI_FOCAL_REF = 715.0873  # bts.py:264
BN_MOMENTUM = 0.01          # bts.py:154 etc.
# A/B switch of the measurement protocol (tools/final_protocol.sh): 0 = every BatchNorm's batch statistics from a bn_stats pass
EPI_STATS = os.environ.get("BTS_EPI_STATS", "1") != "0"
# fused recompute backward of the narrow LPG chains; the tests flip this module constant to train the chains layer by layer
# (the layer-wise path is the checker of the fused one, tests/test_gpu_2_decoder.py)
FUSED_CHAIN_BWD = True
# r6: BatchNorms over a channel concatenation (the dense-ASPP first_bn layers) hand the backward of every input tensor to ONE launch
# per tensor, when all the BatchNorms that normalised it have contributed (DecoderRun._flush_shared_bn); 0 = one bts_bn_bwd per
# BatchNorm over all of its segments, as rounds 3-5 did (the A side of the A/B, and the checker in tests/test_gpu_2_decoder.py)
MULTI_BN_BWD = os.environ.get("BTS_BN_MULTI", "1") != "0"


def reduction_specs(c_in, c_out, is_final):
    """(child name, cin, cout) of the 1x1 convs reduction_1x1.__init__ builds (bts.py:91-108)."""
    out = []
    while c_out >= 4:
        if c_out < 8:
            out.append(("final" if is_final else "plane_params", c_in, 1 if is_final else 3))
            break
        out.append(("inter_%d_%d" % (c_in, c_out), c_in, c_out))
        c_in, c_out = c_out, c_out // 2
    return out


class Act:
    """An activation (NHWC tensor, or a single-channel f32 map [N,H,W]) and its gradient buffer."""
    __slots__ = ("t", "g", "stats", "g_is_dz", "act", "uses")

    def __init__(self, t, act=ACT_NONE):
        self.t, self.g, self.stats = t, None, None
        self.act = act            # activation the producing convolution applied (ACT_ELU: candidates for the derivative fold)
        self.uses = 0             # consumers registered so far (forward order): the FIRST one is the LAST writer of g in backward
        self.g_is_dz = False      # set by the backward op that completed g AND folded this tensor's ELU derivative into it


class DecoderPlan:
    """Static description of the decoder for given encoder channels (bts.__init__, bts.py:149-194)."""

    def __init__(self, feat, nf):
        for c in feat:
            if c % 8:
                raise BtsAmdError("encoder feature channels must be multiples of 8, got %s" % (feat,))
        if nf % 128 or nf < 128:
            raise BtsAmdError("bts_size must be a multiple of 128")
        self.feat, self.nf = list(feat), nf
        L = {}

        def add(name, cout, segs, kk, dil=1, up=False):
            L[name] = ConvLayer(name, cout, segs, kk, dil, up)

        add("upconv5.conv", nf, [feat[4]], 9, up=True)
        add("conv5.0", nf, [nf, feat[3]], 9)
        add("upconv4.conv", nf // 2, [nf], 9, up=True)
        add("conv4.0", nf // 2, [nf // 2, feat[2]], 9)
        cin = {3: nf // 2, 6: nf // 2 + nf // 4 + feat[2], 12: nf + feat[2], 18: nf + nf // 4 + feat[2],
               24: nf + nf // 2 + feat[2]}
        for d in (3, 6, 12, 18, 24):
            add("daspp_%d.atrous_conv.aconv_sequence.1" % d, nf // 2, [cin[d]], 1)
            add("daspp_%d.atrous_conv.aconv_sequence.4" % d, nf // 4, [nf // 2], 9, dil=d)
        self.daspp_cin = cin
        add("daspp_conv.0", nf // 4, [nf // 2] + [nf // 4] * 5, 9)
        self.reduc = {}
        for name, ci, co, fin in (("reduc8x8", nf // 4, nf // 4, False), ("reduc4x4", nf // 4, nf // 8, False),
                                  ("reduc2x2", nf // 8, nf // 16, False), ("reduc1x1", nf // 16, nf // 32, True)):
            chain = []
            for child, a, b in reduction_specs(ci, co, fin):
                key = "%s.reduc.%s" % (name, child + (".0" if child != "plane_params" else ""))
                add(key, b, [a], 1)
                chain.append(key)
            self.reduc[name] = chain
        add("upconv3.conv", nf // 4, [nf // 4], 9, up=True)
        add("conv3.0", nf // 4, [nf // 4, feat[1], 1], 9)
        add("upconv2.conv", nf // 8, [nf // 4], 9, up=True)
        add("conv2.0", nf // 8, [nf // 8, feat[0], 1], 9)
        add("upconv1.conv", nf // 16, [nf // 8], 9, up=True)
        add("conv1.0", nf // 16, [nf // 16, 4], 9)
        add("get_depth.0", 1, [nf // 16], 9)
        self.layers = L
        self.chain_cache = {}
        self.pack_cache = {}


def _cdiv(a, b):
    return (a + b - 1) // b


def _esize(dtype):
    return 4 if dtype == torch.float32 else 2


class PackSet:
    """Packed operands of EVERY conv of the decoder for one (dtype, device): persistent output buffers plus
    device-resident job tables, so a step issues one `bts_pack_weight_batch` for the forward operands, one for the
    data-gradient operands, and one `bts_unpack_wgrad_batch` for all weight gradients (instead of ~170 tiny launches)."""

    def __init__(self, plan, P, dtype):
        self.dtype = dtype
        names = list(plan.layers)
        dev = P[names[0] + ".weight"].device
        self.key = (dtype, tuple(P[n + ".weight"].data_ptr() for n in names))
        self.fwd, self.dgrad = {}, {}
        self.fwd_frag, self.dgrad_frag = {}, {}       # bts_conv_desc_t::w_frag of each operand (ConvLayer.frag_layout)
        self.dwp_off, self.gw_off = {}, {}
        fjobs, djobs, ujobs = [], [], []
        dwp_total = gw_total = 0
        for n in names:
            L = plan.layers[n]
            w = P[n + ".weight"]
            tb = L.tables(dtype, dev)
            ttot = L.nphase * L.T
            lay = L.frag_layout(dtype, L.cout, tb["ktot"])
            if lay:       # MFMA A-fragment order for conv_igemm_res: zeroed once, the pack kernel writes the real entries
                out = torch.zeros(L.frag_bytes(L.cout, tb["ktot"], False) // 2, dtype=dtype, device=dev)
            else:
                out = torch.empty((L.cout, ttot, tb["ktot"]), dtype=dtype, device=dev)
            self.fwd[n], self.fwd_frag[n] = out, lay
            fjobs.append(self._pjob(w, out, tb["cmap"], L, 0, L.cout, tb["ktot"], ttot, lay, L._launch_taps(False)[1]))
            for i, rows in enumerate(tb["seg_rows"]):
                lay = L.frag_layout(dtype, rows.numel(), tb["cout_pad"], True)
                if lay:
                    o = torch.zeros(L.frag_bytes(rows.numel(), tb["cout_pad"], True) // 2, dtype=dtype, device=dev)
                else:
                    o = torch.empty((rows.numel(), ttot, tb["cout_pad"]), dtype=dtype, device=dev)
                self.dgrad[(n, i)], self.dgrad_frag[(n, i)] = o, lay
                djobs.append(self._pjob(w, o, rows, L, 1, rows.numel(), tb["cout_pad"], ttot, lay, L._launch_taps(True)[1]))
            self.dwp_off[n] = (dwp_total, (L.cout, ttot, tb["ktot"]))
            self.gw_off[n] = (gw_total, tuple(w.shape))
            uj = _lib.UnpackJob()
            uj.dwp_off, uj.gw_off, uj.kinv = dwp_total, gw_total, tb["kinv"].data_ptr()
            uj.Cout, uj.Cin, uj.KK, uj.K, uj.T = L.cout, L.cin, L.kk, tb["ktot"], ttot
            for t, m in enumerate(L.masks):
                uj.tapmask[t] = m
            ujobs.append(uj)
            dwp_total += L.cout * ttot * tb["ktot"]
            gw_total += w.numel()
        self.dwp_total, self.gw_total = dwp_total, gw_total
        # block ranges (include/bts_amd.h: 32x32 (co, ci) tiles per pack job, 256 (co, ci) pairs per unpack block)
        self.fblocks = self._assign(fjobs, lambda j: _cdiv(j.R if j.mode == 0 else j.K, 32) * _cdiv(j.K if j.mode == 0 else j.R, 32))
        self.dblocks = self._assign(djobs, lambda j: _cdiv(j.R if j.mode == 0 else j.K, 32) * _cdiv(j.K if j.mode == 0 else j.R, 32))
        self.ublocks = self._assign(ujobs, lambda j: _cdiv(j.Cout * j.Cin, 256))
        self.fjobs, self.nf = self._upload(fjobs, dev), len(fjobs)
        self.djobs, self.nd = self._upload(djobs, dev), len(djobs)
        self.ujobs, self.nu = self._upload(ujobs, dev), len(ujobs)

    @staticmethod
    def _assign(jobs, nblocks):
        total = 0
        for j in jobs:
            j.first_block = total
            total += nblocks(j)
        return total

    @staticmethod
    def _pjob(w, out, cmap, L, mode, R, K, ttot, layout=0, tp=0):
        j = _lib.PackJob()
        j.w, j.out, j.cmap = w.data_ptr(), out.data_ptr(), cmap.data_ptr()
        j.Cout, j.Cin, j.KK, j.mode, j.R, j.K, j.T = L.cout, L.cin, L.kk, mode, R, K, ttot
        j.layout, j.Tp = layout, (tp or ttot)
        for t, m in enumerate(L.masks):
            j.tapmask[t] = m
        return j

    @staticmethod
    def _upload(jobs, dev):
        raw = b"".join(bytes(j) for j in jobs)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def pack_forward(self):
        if profiler.ACTIVE is not None:
            profiler.note("pack_weight_batch", "hbm", self.gw_total * 4 + sum(t.numel() for t in self.fwd.values()) * _esize(self.dtype))
        _lib.call("bts_pack_weight_batch", C.c_void_p(self.fjobs.data_ptr()), self.nf, self.fblocks, _lib.dtype_code(self.dtype),
                  _lib.stream_ptr())

    def pack_dgrad(self):
        if profiler.ACTIVE is not None:
            profiler.note("pack_weight_batch", "hbm", self.gw_total * 4 + sum(t.numel() for t in self.dgrad.values()) * _esize(self.dtype))
        _lib.call("bts_pack_weight_batch", C.c_void_p(self.djobs.data_ptr()), self.nd, self.dblocks, _lib.dtype_code(self.dtype),
                  _lib.stream_ptr())

    def unpack_all(self, dwp_arena, gw_arena):
        if profiler.ACTIVE is not None:
            profiler.note("unpack_wgrad_batch", "hbm", (self.dwp_total + self.gw_total) * 4)
        _lib.call("bts_unpack_wgrad_batch", C.c_void_p(self.ujobs.data_ptr()), self.nu, self.ublocks,
                  C.c_void_p(dwp_arena.data_ptr()), C.c_void_p(gw_arena.data_ptr()), _lib.stream_ptr())


class DecoderRun:
    """One forward (and optionally backward) pass."""

    def __init__(self, plan, P, bn_training, max_depth, dataset, dtype, record):
        self.plan, self.P, self.bn_training = plan, P, bn_training
        self.max_depth, self.dataset, self.dtype, self.record = float(max_depth), dataset, dtype, record
        self.v = vec_of(dtype)
        self.tape, self.grads = [], {}
        self.feat_acts, self.feat_src = [], []
        self.outs = None
        names = list(plan.layers)
        key = (dtype, tuple(P[n + ".weight"].data_ptr() for n in names))
        dev = P[names[0] + ".weight"].device
        ps = plan.pack_cache.get((dtype, dev))           # per device: DataParallel replicas share the plan
        if ps is None or ps.key != key:
            ps = PackSet(plan, P, dtype)
            plan.pack_cache[(dtype, dev)] = ps
        self.packs = ps
        self.dwp_arena = None
        self.wgrad_pending = []
        self.bn_shared = {}          # id(Act) -> pending backward contributions of the concatenation BatchNorms that normalised it

    # ---- ops -------------------------------------------------------------------------------
    def _use(self, a, can_fold):
        """Register a consumer of activation `a` (call once per consumer, in forward order).  Returns True when this consumer
        should fold a's ELU derivative into the gradient it writes: it is a's FIRST consumer -- the tape runs backwards, so its
        backward is the LAST writer of a.g, i.e. the one that completes it -- a is a conv+ELU output, and the consumer's kernel
        has the fold (convolution data-gradients, the fused chain backward, the BatchNorm backward)."""
        first = a.uses == 0
        a.uses += 1
        return bool(self.record and first and can_fold and a.act == ACT_ELU and a.t.dim() == 4)

    def feature(self, f, relu=False):
        """Encoder feature (NCHW, f32 or bf16 under autocast) -> NHWC activation; no intermediate casts."""
        src = f.detach()
        if src.dtype not in (torch.float32, torch.bfloat16):
            src = src.float()
        nhwc = src.permute(0, 2, 3, 1)
        if (not src.is_contiguous() and nhwc.is_contiguous() and src.dtype == self.dtype and src.shape[1] % self.v == 0
                and src.data_ptr() % 16 == 0):
            # a channels-last producer (encoder run in torch.channels_last): the tensor already IS the decoder's layout.  A skip
            # feature is used in place (read-only; `_version` is re-checked in backward), the dense feature takes its ReLU
            # (bts.py:198) in one same-layout pass; gradients go back as channels-last views -- no transposes either way.
            a = Act(ops.affine_act(nhwc, None, None, ACT_RELU) if relu else nhwc)
            self.feat_acts.append(a)
            self.feat_src.append(("cl", relu, f, f._version))
            return a
        if not src.is_contiguous():
            src = src.contiguous()
        a = Act(ops.nchw_to_nhwc(src, self.dtype, relu))
        self.feat_acts.append(a)
        self.feat_src.append((src if relu else None, f.shape[1], src.dtype))
        return a

    def conv(self, name, segs, act, out_map=False, out_f32=False, out_scale=1.0, out_scale_n=None, stats_for=None):
        """stats_for: prefix of a BatchNorm that normalises this output (bts.py:57-62, 199-208, 231-232).  When that BatchNorm is in
        training mode the convolution's epilogue also forms the output's batch statistics where its kernel can (ConvLayer.forward:
        stats), so bn_cat() finds them on the activation and needs no pass of its own over the tensor."""
        L = self.plan.layers[name]
        wp = self.packs.fwd[name]
        if len({id(s) for s in segs}) != len(segs):       # the ELU-fold bookkeeping (one fold per Act) relies on it
            raise BtsAmdError("conv %s: the same activation passed as two input segments" % name)
        x = [s.t for s in segs]
        N, Hx, Wx, _ = x[0].shape
        Ho, Wo = (2 * Hx, 2 * Wx) if L.up else (Hx, Wx)
        dev = x[0].device
        if out_map:
            out = torch.empty((N, Ho, Wo), dtype=torch.float32, device=dev)
        else:
            odt = torch.float32 if out_f32 else self.dtype
            cp = pad_to(L.cout, vec_of(odt))
            out = (torch.zeros if cp != L.cout else torch.empty)((N, Ho, Wo, cp), dtype=odt, device=dev)
        st = [] if (stats_for is not None and EPI_STATS and self.bn_training[stats_for] and not out_map and not out_f32) else None
        L.forward(x, wp, out, act, out_scale, out_scale_n, self.packs.fwd_frag[name], st)
        y = Act(out, act if (not out_map and not out_f32 and out_scale == 1.0 and out_scale_n is None) else ACT_NONE)
        if st:
            y.stats = st[0]
        folds = [self._use(s, s.t.dtype == self.dtype) for s in segs]
        if self.record:
            def bwd():
                if y.g is None:
                    return
                if out_map:
                    dz = ops.act_bwd(y.g, y.t, act, out_dtype=self.dtype, out_channels=self.v, y_scale=out_scale,
                                     y_scale_n=out_scale_n)
                elif act == ACT_NONE or y.g_is_dz:
                    dz = y.g
                else:
                    dz = ops.act_bwd(y.g, y.t, act, out=y.g)
                accs = []
                for s in segs:
                    accs.append(s.g is not None)
                    if s.g is None:
                        s.g = torch.empty(s.t.shape, dtype=self.dtype, device=dev)
                done = set()
                if len(segs) == 2 and not folds[1] and L.dual_dgrad_ok(dz.dtype, 0, 1):
                    # both data gradients from one pass over dz (conv1: towards upconv1 and towards the depth-map slots); a layout
                    # outside the form's domain (the library answers BTS_ERR_UNSUPPORTED before launching anything) takes two launches
                    try:
                        L.dgrad_dual(dz, self.packs.dgrad[(name, 0)], 0, segs[0].g, accs[0], segs[0].t if folds[0] else None,
                                     self.packs.dgrad[(name, 1)], 1, segs[1].g, accs[1])
                        done = {0, 1}
                    except BtsAmdError as e:
                        if e.code != _lib.ERR_UNSUPPORTED:
                            raise
                for i, s in enumerate(segs):
                    if i not in done:
                        L.dgrad(dz, self.packs.dgrad[(name, i)], i, s.g, accs[i], s.t if folds[i] else None, self.packs.dgrad_frag[(name, i)])
                    if folds[i]:
                        s.g_is_dz = True
                off, shape = self.packs.dwp_off[name]
                n_el = shape[0] * shape[1] * shape[2]
                self._wgrad(L, x, dz, self.dwp_arena[off:off + n_el].view(shape))
            self.tape.append(bwd)
        return y

    # The weight gradients of the SMALL ring-kernel layers (the dense ASPP, reduc8x8's 128 -> 128 layer) are deferred and leave five at a
    # time, in the order the backward pass reaches them (bts_conv_wgrad_group): a weight gradient depends only on its own layer's
    # (dz, inputs) -- both stay alive until the pass ends -- and a grouped launch pays the split-K atomics and the pipeline prologue
    # once per group instead of once per layer.  Arrival order interleaves the ASPP's 1x1 layers (bound by fabric traffic: 87 FLOP per
    # staged byte with little reuse between workgroups) with its dilated 3x3 layers (MFMA-bound, L2-friendly), which is what pays:
    # eleven launches, 637 us -> 170 + 208 + 39 us (gpurun r04k); groups of like layers -- five 3x3, five 1x1 -- measured 277 + 262 us,
    # groups of six and five 288 + 248 (profiles/r04_wgrad_group_*.json).
    WGRAD_GROUP = 5

    def _wgrad(self, L, x, dz, dwp):
        N, Hx, Wx, _ = x[0].shape
        if dz.dtype != x[0].dtype or not L.wgrad_groupable(x[0].dtype, N, Hx, Wx):      # keyed like _wgrad_desc: the inputs' dtype
            L.wgrad_packed(x, dz, dwp)
            return
        self.wgrad_pending.append((L, x, dz, dwp))
        if len(self.wgrad_pending) == self.WGRAD_GROUP:
            self._flush_wgrads()

    def _flush_wgrads(self):
        items, self.wgrad_pending = self.wgrad_pending, []
        if len(items) == 1:
            L, x, dz, dwp = items[0]
            L.wgrad_packed(x, dz, dwp)
        elif items:
            try:
                ConvLayer.wgrad_group(items)
            except BtsAmdError as e:           # outside the grouped form's domain (nothing was launched): layer by layer
                if e.code != _lib.ERR_UNSUPPORTED:
                    raise
                for L, x, dz, dwp in items:
                    L.wgrad_packed(x, dz, dwp)

    def conv_c1(self, name, x, out_scale, out_scale_n):
        """3x3 convolution to one channel + sigmoid * scale (get_depth, bts.py:193-194, 262-264) on the streaming kernels of
        csrc/conv_c1.hip: forward, data gradient and weight gradient (the sigmoid derivative is formed inside the two backward
        kernels; no dz map exists).  Falls back to the generic convolution outside their domain."""
        L = self.plan.layers[name]
        w = self.P[name + ".weight"]
        # the streaming kernels index w[c * 9 + tap] for every c below x's PADDED channel count: only when no channel is padding
        if not (L.cout == 1 and L.kk == 9 and L.dil == 1 and not L.up and ops.conv_c1_supported(x.t, w)):
            return self.conv(name, [x], ACT_SIGMOID, out_map=True, out_scale=out_scale, out_scale_n=out_scale_n)
        y = Act(ops.conv3x3_c1_fwd(x.t, w, out_scale, out_scale_n))
        fold = self._use(x, True)
        if self.record:
            def bwd():
                if y.g is None:
                    return
                acc = x.g is not None
                if not acc:
                    x.g = torch.empty(x.t.shape, dtype=self.dtype, device=x.t.device)
                ops.conv3x3_c1_dgrad(y.g, y.t, w, x.g, acc, out_scale, out_scale_n, x.t if fold else None)
                if fold:
                    x.g_is_dz = True
                off, shape = self.packs.dwp_off[name]
                # (one fused pass over x for both gradients was built and measured in round 4: 227 us against 110 + 106 -- its 144
                # accumulators + weights per lane leave two waves per SIMD; removed)
                ops.conv3x3_c1_wgrad(y.g, y.t, x.t, self.dwp_arena[off:off + shape[0] * shape[1] * shape[2]].view(shape),
                                     out_scale, out_scale_n)
            self.tape.append(bwd)
        return y

    def bn(self, x, prefix, eps, relu=False, x_act=ACT_NONE, relu_copy=False):
        return self.bn_cat([x], prefix, eps, relu, x_act, relu_copy)

    def bn_cat(self, segs, prefix, eps, relu, x_act=ACT_NONE, relu_copy=False):
        """BN(+ReLU) over the channel concatenation of `segs`, materialised as one tensor by ONE launch (csrc/elementwise.hip,
        bn_apply_ms_kernel); its backward is one reduction + one apply launch for all segments.

        x_act = ACT_ELU: the (single) input is the ELU output of a convolution that feeds nothing else (bts.py:199-208: upconv5,
        upconv4, conv4, upconv3, upconv2); the backward then folds the ELU derivative into the gradient it writes, and the
        producing convolution's backward starts from it (no separate activation-derivative pass).
        relu_copy: also return relu(output) from the same launch (bts.py:208-210: bn4_2's output feeds daspp_conv as is and
        daspp_3 through a ReLU)."""
        P = self.P
        N, H, W, _ = segs[0].t.shape
        ctot = sum(s.t.shape[3] for s in segs)
        dev = segs[0].t.device
        train = self.bn_training[prefix]
        g, b, rm, rv = P[prefix + ".weight"], P[prefix + ".bias"], P[prefix + ".running_mean"], P[prefix + ".running_var"]
        if g.numel() != ctot:
            raise BtsAmdError("BatchNorm %s has %d channels, its input %d" % (prefix, g.numel(), ctot))
        stats, c0 = [], 0
        for s in segs:
            Cc = s.t.shape[3]
            if train:
                if s.stats is None:
                    s.stats = ops.bn_stats(s.t)      # shared by every BN that sees this tensor
                stats.append(s.stats)
            else:
                stats.append((rm[c0:c0 + Cc], rv[c0:c0 + Cc]))
            c0 += Cc
        out = torch.empty((N, H, W, ctot), dtype=self.dtype, device=dev)
        out2 = torch.empty_like(out) if relu_copy else None
        ops.bn_apply([s.t for s in segs], stats, g, b, eps, relu, out, out2, BN_MOMENTUM if train else 0.0,
                     rm if train else None, rv if train else None, tag=prefix)
        if train:
            nbt = P.get(prefix + ".num_batches_tracked")
            if nbt is not None:
                nbt.add_(1)
        y = Act(out)
        y2 = Act(out2) if relu_copy else None
        firsts = [self._use(sg, True) for sg in segs]
        fold = x_act == ACT_ELU and len(segs) == 1 and firsts[0]
        if x_act != ACT_NONE and len(segs) != 1:
            raise BtsAmdError("bn_cat: x_act is only supported for a single ELU input")
        # A concatenation BatchNorm in training mode shares each input tensor (and its batch statistics) with the other BatchNorms that
        # see it: the backward of a tensor waits until the LAST of them (in backward order) has contributed and then runs once
        # (_flush_shared_bn).  A tensor's producer comes before all of its consumers in the schedule, so its backward runs after that.
        shared = MULTI_BN_BWD and self.record and train and len(segs) > 1
        if shared:
            for sg in segs:
                self.bn_shared.setdefault(id(sg), {"act": sg, "left": 0, "items": []})["left"] += 1
        if self.record:
            def bwd():
                if shared:
                    sums = torch.zeros((2, ctot), dtype=torch.float32, device=dev) if y.g is None else \
                        torch.empty((2, ctot), dtype=torch.float32, device=dev)
                    self.grads[prefix + ".weight"], self.grads[prefix + ".bias"] = sums[1], sums[0]
                    c0, ready = 0, []
                    for sg, st in zip(segs, stats):
                        Cc = sg.t.shape[3]
                        ent = self.bn_shared[id(sg)]
                        if y.g is not None:
                            ent["items"].append((y.g[..., c0:c0 + Cc], g[c0:c0 + Cc], b[c0:c0 + Cc], sums[0, c0:c0 + Cc], sums[1, c0:c0 + Cc],
                                                 float(eps), bool(relu), st, prefix))
                        ent["left"] -= 1
                        if ent["left"] == 0 or len(ent["items"]) == _lib.BN_MAX_MULTI:
                            ready.append(ent)
                        c0 += Cc
                    self._flush_shared_bn(ready)
                    return
                if y.g is None:
                    return
                accs = []
                for s in segs:
                    accs.append(s.g is not None)
                    if s.g is None:
                        s.g = torch.empty(s.t.shape, dtype=self.dtype, device=dev)
                if fold:
                    if accs[0]:
                        raise BtsAmdError("bn_cat: ELU-folded input of %s has another consumer" % prefix)
                    segs[0].g_is_dz = True
                db, dg = ops.bn_bwd_ms(y.g, [s.t for s in segs], [s.g for s in segs], accs, stats, g, b, eps, relu, train, fold, tag=prefix)
                self.grads[prefix + ".weight"] = dg
                self.grads[prefix + ".bias"] = db
            self.tape.append(bwd)
            if relu_copy:
                def bwd2():
                    if y2.g is None:
                        return
                    if y.g is None:
                        y.g = ops.act_bwd(y2.g, y2.t, ACT_RELU)
                    else:
                        ops.act_bwd(y2.g, y2.t, ACT_RELU, out=y.g, accumulate=True)
                self.tape.append(bwd2)
        return (y, y2) if relu_copy else y

    def _flush_shared_bn(self, ready):
        """Backward of the input tensors whose concatenation BatchNorms have all contributed: tensors that were normalised by the SAME
        BatchNorms (same count, eps, ReLU) share one bts_bn_bwd_multi launch (up to three tensors; it reads each once per pass for all
        of its BatchNorms and writes its gradient once)."""
        groups = {}
        for ent in ready:
            items, ent["items"] = ent["items"], []
            if not items:
                continue
            sg = ent["act"]
            for key in sorted({(it[5], it[6]) for it in items}):
                group = [it for it in items if (it[5], it[6]) == key]
                acc = sg.g is not None
                if not acc:
                    sg.g = torch.empty(sg.t.shape, dtype=self.dtype, device=sg.t.device)
                mean, var = group[0][7]
                tag = "+".join(it[8].split(".")[0] for it in group)
                groups.setdefault((key, len(group), tag), []).append((sg.t, mean, var, sg.g, acc, [it[:5] for it in group]))
        for ((eps, relu), n, tag), tensors in groups.items():
            for i in range(0, len(tensors), _lib.BN_MULTI_TENSORS):
                ops.bn_bwd_multi(tensors[i:i + _lib.BN_MULTI_TENSORS], eps, relu, True, tag=tag)

    def relu(self, x):
        self._use(x, False)
        y = Act(ops.affine_act(x.t, None, None, ACT_RELU))
        if self.record:
            def bwd():
                if y.g is None:
                    return
                if x.g is None:
                    x.g = ops.act_bwd(y.g, y.t, ACT_RELU)
                else:
                    ops.act_bwd(y.g, y.t, ACT_RELU, out=x.g, accumulate=True)
            self.tape.append(bwd)
        return y

    def head(self, raw, k):
        self._use(raw, False)
        d = Act(ops.lpg_head_fwd(raw.t, k, self.max_depth))
        if self.record:
            def bwd():
                if d.g is None:
                    return
                raw.g = ops.lpg_head_bwd(raw.t, d.g, k, self.max_depth, self.dtype, self.v)
            self.tape.append(bwd)
        return d

    def plane(self, raw):
        """(theta, phi, dist) -> un-normalised plane parameters (bts.py:112-120) as its own op: the standalone reduction_1x1
        module's tail; inside the decoder this arithmetic is part of the fused head kernels."""
        self._use(raw, False)
        y = Act(ops.plane_fwd(raw.t, self.max_depth))
        if self.record:
            def bwd():
                if y.g is None:
                    return
                raw.g = ops.plane_bwd(raw.t, y.g, self.max_depth, self.dtype, self.v)
            self.tape.append(bwd)
        return y

    def slots(self, maps, ds, N, H, W):
        for m in maps:
            self._use(m, False)
        y = Act(ops.pack_maps([m.t for m in maps], ds, N, H, W, self.dtype))
        if self.record:
            def bwd():
                if y.g is None:
                    return
                for m in maps:
                    if m.g is None:
                        m.g = torch.zeros_like(m.t)
                ops.unpack_maps(y.g, [m.g for m in maps], ds)
            self.tape.append(bwd)
        return y

    def _chain_pack(self, name, start, ws, with_t):
        """Fragment buffers of the chain layers ws (= layers start.. of chain `name`), repacked from the live weights
        on every pass (two gathers through cached indices): nothing keyed on tensor versions, so in-place /
        graph-replayed optimizer updates are always seen."""
        cache = self.plan.chain_cache
        key = (name, start, self.dtype, ws[0].device)
        pk = cache.get(key)
        if pk is None:
            pk = chain_mod.ChainPacker([(w.shape[0], w.shape[1]) for w in ws], self.dtype, ws[0].device)
            cache[key] = pk
        return pk.pack(ws, with_t)

    def chain_fused(self, name, x, k):
        """reduction_1x1 chain (+ plane head + LPG for k > 1) in one kernel (csrc/lpg_chain.hip).  Without recording:
        the whole chain, any instantiated shape.  With recording: the longest halving tail that also has the fused
        recompute backward (bf16, <= 64 channels) -- all of reduc2x2 / reduc1x1, and reduc8x8 / reduc4x4 behind their
        128-channel layers, which run layer by layer; the fused part stores nothing but its input.  None if no
        instantiation applies."""
        keys = self.plan.reduc[name]
        ws = [self.P[key + ".weight"] for key in keys]
        train = self.record
        start = None
        for s0 in range(len(keys) - 1):
            c0, same = ws[s0].shape[1], ws[s0].shape[0] == ws[s0].shape[1]
            ok = chain_mod.supported(c0, same, k)
            if train:
                ok = ok and FUSED_CHAIN_BWD and chain_mod.bwd_supported(c0, same, k, self.dtype)
            if ok:
                start = s0
                break
            if not train:
                break                                   # no-grad: whole chain or nothing
        if start is None or (start == 0 and x.t.shape[3] != ws[0].shape[1]):
            return None
        for key in keys[:start]:
            x = self.conv(key, [x], ACT_ELU)                       # bts.py:101-105, the wide layers
        keys, ws = keys[start:], ws[start:]
        c0, same = ws[0].shape[1], ws[0].shape[0] == ws[0].shape[1]
        if x.t.shape[3] != c0:
            raise BtsAmdError("reduction chain %s: unexpected channel padding" % name)
        frags, frags_t = self._chain_pack(name, start, ws, train)
        fold = self._use(x, train)              # the recompute backward re-reads x anyway: folding x's ELU derivative is free
        d = Act(chain_mod.chain_fwd(x.t, frags, c0, same, k, self.max_depth))
        if train:
            def bwd():
                if d.g is None:
                    return
                acc = x.g is not None
                if not acc:
                    x.g = torch.empty(x.t.shape, dtype=self.dtype, device=x.t.device)
                gws = []
                for key in keys:
                    off, shape = self.packs.dwp_off[key]
                    gws.append(self.dwp_arena[off:off + shape[0] * shape[1] * shape[2]].view(shape[0], shape[1] * shape[2]))
                chain_mod.chain_bwd(x.t, frags, frags_t, c0, k, self.max_depth, d.g, x.g, acc, gws, fold)
                if fold:
                    x.g_is_dz = True
            self.tape.append(bwd)
        return d

    def lpg_branch(self, name, x, k):
        """reduction chain + LPG head: one fused kernel where an instantiation exists, layer-wise otherwise."""
        d = self.chain_fused(name, x, k)
        if d is not None:
            return d
        y = self.chain(name, x)
        return y if k == 1 else self.head(y, k)

    def chain(self, name, x):
        keys = self.plan.reduc[name]
        for key in keys[:-1]:
            x = self.conv(key, [x], ACT_ELU)                       # bts.py:101-105
        if name == "reduc1x1":
            return self.conv(keys[-1], [x], ACT_SIGMOID, out_map=True)   # bts.py:93-96
        return self.conv(keys[-1], [x], ACT_NONE, out_f32=True)          # bts.py:98-99 (raw plane params)

    def atrous(self, d, x):
        """atrous_conv minus first_bn (bts.py:57-62): x is already BN'd/ReLU'd."""
        p = "daspp_%d.atrous_conv.aconv_sequence" % d
        a = self.conv(p + ".1", [x], ACT_NONE, stats_for=p + ".2")
        a = self.bn(a, p + ".2", 1e-5, relu=True)                 # default eps (bts.py:60)
        nxt = {3: 6, 6: 12, 12: 18, 18: 24}.get(d)                # the next block's first_bn sees this output (bts.py:211-218)
        return self.conv(p + ".4", [a], ACT_NONE, stats_for="daspp_%d.atrous_conv.first_bn" % nxt if nxt else None)

    # ---- schedule (bts.forward, bts.py:196-266) ------------------------------------------------
    def forward(self, features, focal):
        f = features
        self.packs.pack_forward()
        N, _, H2, W2 = f[0].shape
        H, W = 2 * H2, 2 * W2
        s0, s1, s2, s3 = (self.feature(f[i]) for i in range(4))
        dense = self.feature(f[4], relu=True)                               # :198
        u5 = self.bn(self.conv("upconv5.conv", [dense], ACT_ELU, stats_for="bn5"), "bn5", 1.1e-5, x_act=ACT_ELU)      # :199-200
        i5 = self.conv("conv5.0", [u5, s3], ACT_ELU)                        # :201-202
        u4 = self.bn(self.conv("upconv4.conv", [i5], ACT_ELU, stats_for="bn4"), "bn4", 1.1e-5, x_act=ACT_ELU)        # :204-205
        i4, i4r = self.bn(self.conv("conv4.0", [u4, s2], ACT_ELU, stats_for="bn4_2"), "bn4_2", 1.1e-5, x_act=ACT_ELU, relu_copy=True)   # :206-208
        d3 = self.atrous(3, i4r)                                            # :210 (no first_bn; the ReLU of :58 comes with bn4_2)
        cat = [u4, s2, d3]
        dk = {3: d3}
        for d in (6, 12, 18, 24):                                           # :211-218
            n = self.bn_cat(cat, "daspp_%d.atrous_conv.first_bn" % d, 1.1e-5, relu=True)
            dk[d] = self.atrous(d, n)
            cat = cat + [dk[d]]
        df = self.conv("daspp_conv.0", [i4, dk[3], dk[6], dk[12], dk[18], dk[24]], ACT_ELU)   # :219-220

        d8 = self.lpg_branch("reduc8x8", df, 8)                             # :222-228
        u3 = self.bn(self.conv("upconv3.conv", [df], ACT_ELU, stats_for="bn3"), "bn3", 1.1e-5, x_act=ACT_ELU)        # :231-232
        i3 = self.conv("conv3.0", [u3, s1, self.slots([d8], [4], N, H // 4, W // 4)], ACT_ELU)   # :229, 233-234
        d4 = self.lpg_branch("reduc4x4", i3, 4)                             # :236-242
        u2 = self.bn(self.conv("upconv2.conv", [i3], ACT_ELU, stats_for="bn2"), "bn2", 1.1e-5, x_act=ACT_ELU)        # :245-246
        i2 = self.conv("conv2.0", [u2, s0, self.slots([d4], [2], N, H // 2, W // 2)], ACT_ELU)   # :243, 247-248
        d2 = self.lpg_branch("reduc2x2", i2, 2)                             # :250-256
        u1 = self.conv("upconv1.conv", [i2], ACT_ELU)                       # :258
        r1 = self.lpg_branch("reduc1x1", u1, 1)                             # :259
        i1 = self.conv("conv1.0", [u1, self.slots([r1, d2, d4, d8], [1, 1, 1, 1], N, H, W)], ACT_ELU)   # :260-261
        scale_n = None
        if self.dataset == "kitti":                                         # :263-264
            scale_n = (focal.detach().to(device=i1.t.device, dtype=torch.float32) / KITTI_FOCAL_REF).contiguous()
        depth = self.conv_c1("get_depth.0", i1, self.max_depth, scale_n)
        self.outs = [d8, d4, d2, r1, depth]
        return tuple(o.t.view(N, 1, H, W) for o in self.outs)

    def backward(self, grad_outs):
        if not self.record:
            raise BtsAmdError("backward() on a decoder pass that was run without recording")
        for o, g in zip(self.outs, grad_outs):
            if g is not None:
                # the LPG maps also receive gradient through conv1/conv3/conv2: own the buffer
                o.g = g.reshape(o.t.shape).to(torch.float32).clone(memory_format=torch.contiguous_format)
        dev = next(iter(self.packs.fwd.values())).device
        self.packs.pack_dgrad()
        self.dwp_arena = torch.zeros(self.packs.dwp_total, dtype=torch.float32, device=dev)     # one memset for every layer
        try:
            for fn in reversed(self.tape):
                fn()
            self._flush_wgrads()
            if any(e["items"] or e["left"] for e in self.bn_shared.values()):
                raise BtsAmdError("a concatenation BatchNorm's backward was left pending at the end of the pass")
        finally:
            # an error mid-pass must not leave (layer, inputs, dz, slot-of-THIS-arena) tuples behind for a retried backward to flush
            self.wgrad_pending = []
        self.tape = []
        gw_arena = torch.empty(self.packs.gw_total, dtype=torch.float32, device=dev)
        self.packs.unpack_all(self.dwp_arena, gw_arena)
        for name, (off, shape) in self.packs.gw_off.items():
            n_el = 1
            for d in shape:
                n_el *= d
            self.grads[name + ".weight"] = gw_arena[off:off + n_el].view(shape)
        self.dwp_arena = None
        gfeats = []
        for a, src in zip(self.feat_acts, self.feat_src):
            if src[0] == "cl":                               # channels-last feature (feature()): gradient returned as a view
                _, relu, f, version = src
                if not relu and f._version != version:
                    raise BtsAmdError("a channels-last encoder feature was modified in place between the decoder's forward and "
                                      "backward (the decoder reads it without a copy)")
                g = a.g
                if g is not None and relu:
                    g = ops.act_bwd(g, a.t, ACT_RELU, out=g)
                gfeats.append(None if g is None else g.permute(0, 3, 1, 2))
                continue
            relu_src, Cc, sdt = src
            gfeats.append(None if a.g is None else ops.nhwc_to_nchw(a.g, Cc, relu_src, out_dtype=sdt))
        # feature() was called for skips 0..3 then the dense map
        return gfeats, self.grads
