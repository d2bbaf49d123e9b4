"""Drop-in replacement for the names exported by the reference's ``pytorch/bts.py``.

Same public surface: ``BtsModel(params)``, ``bts`` (decoder), ``encoder``,
``silog_loss``, ``weights_init_xavier``, ``bn_init_as_tf``, plus the building-block classes
``atrous_conv``, ``upconv``, ``reduction_1x1``, ``local_planar_guidance`` -- same constructor
arguments, same module/parameter names and registration order (bts.py:149-194), hence the same
``state_dict`` keys and AdamW index-keyed optimizer state, so checkpoints written by either
implementation load in the other (bts_main.py:376-397, 498-503).

The difference is what ``forward`` does: the decoder, LPG and the loss execute hand-written
gfx950 kernels through the C ABI (bts_amd/_lib.py); the nn.Conv2d / nn.BatchNorm2d objects
below are parameter containers (they keep ``isinstance`` based code such as
``weights_init_xavier`` / ``bn_init_as_tf`` / ``set_misc`` working) and are never called.
There is no CPU execution path: calling ``forward`` without the library or without a HIP
device raises.
"""

import operator

import torch
import torch.nn as nn


from . import ops
from ._lib import ACT_ELU, ACT_NONE, BtsAmdError, require_gpu
from .conv import ConvLayer
from .decoder import DecoderPlan, DecoderRun, reduction_specs

__all__ = ["BtsModel", "bts", "encoder", "silog_loss", "weights_init_xavier", "bn_init_as_tf",
           "atrous_conv", "upconv", "reduction_1x1", "local_planar_guidance"]


# ---------------------------------------------------------------------------------------------
# init helpers (bts.py:26-38) -- plain PyTorch, operate on the parameter containers
# ---------------------------------------------------------------------------------------------
def bn_init_as_tf(m):
    """bts.py:26-31: freeze a BatchNorm layer to inference behaviour (TF is_training=False)."""
    if isinstance(m, nn.BatchNorm2d):
        m.track_running_stats = True
        m.eval()
        m.affine = True
        m.requires_grad = True


def weights_init_xavier(m):
    """bts.py:33-38."""
    if isinstance(m, nn.Conv2d):
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            torch.nn.init.zeros_(m.bias)


# ---------------------------------------------------------------------------------------------
# silog loss (bts.py:41-48)
# ---------------------------------------------------------------------------------------------
class _SilogFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, gt, mask, vf):
        require_gpu(est)
        est_c = est.detach().to(torch.float32).contiguous()
        gt_c = gt.detach().to(torch.float32).contiguous()
        m = mask.detach().contiguous()
        if m.dtype != torch.bool and m.dtype != torch.uint8:
            m = m != 0
        if m.shape != est_c.shape:
            m = m.expand_as(est_c).contiguous()
        loss, stats = ops.silog_fwd(est_c, gt_c, m, vf)
        ctx.save_for_backward(est_c, gt_c, m, stats, loss)
        ctx.vf = vf
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        est, gt, m, stats, loss = ctx.saved_tensors
        g = ops.silog_bwd(est, gt, m, ctx.vf, stats, loss, gloss.reshape(1).to(torch.float32).contiguous())
        return g, None, None, None


class silog_loss(nn.Module):
    def __init__(self, variance_focus):
        super().__init__()
        self.variance_focus = variance_focus

    def forward(self, depth_est, depth_gt, mask):
        return _SilogFn.apply(depth_est, depth_gt, mask, float(self.variance_focus))


# ---------------------------------------------------------------------------------------------
# building blocks: parameter containers with the reference's names; standalone forward() of the
# LPG layer goes through the LPG op of the C ABI
# ---------------------------------------------------------------------------------------------
class _BlockPlan:
    """Duck-typed DecoderPlan of ONE building block (what DecoderRun / PackSet need: layers, reduc, caches)."""

    def __init__(self, layers, reduc=None):
        self.layers, self.reduc = layers, reduc or {}
        self.chain_cache, self.pack_cache = {}, {}


class _BlockFn(torch.autograd.Function):
    """Standalone forward / backward of a building block on the decoder's executor (same kernels, same tape): the
    reference exposes atrous_conv / upconv / reduction_1x1 as working modules (bts.py:51-122)."""

    @staticmethod
    def forward(ctx, mod, names, record, x, *params):
        require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        for k, b in mod.named_buffers():
            P[k] = b
        bn_training = {k: m.training for k, m in mod.named_modules() if isinstance(m, nn.BatchNorm2d)}
        run = DecoderRun(mod._plan, P, bn_training, float(getattr(mod, "max_depth", 1.0)), "block", mod.compute_dtype, record)
        run.packs.pack_forward()
        y = mod._schedule(run, run.feature(x))
        run.outs = []
        ctx.run, ctx.y, ctx.names = run, y, names
        if y.t.dim() == 3:                                       # single-channel f32 map (reduction_1x1 final)
            return y.t.unsqueeze(1).clone()
        return ops.nhwc_to_nchw(y.t, mod._cout, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, gy):
        run, y = ctx.run, ctx.y
        if run is None:
            raise BtsAmdError("one backward tape per forward pass: run the block's forward again")
        if y.t.dim() == 3:
            y.g = gy.reshape(y.t.shape).to(torch.float32).contiguous().clone()
        else:
            y.g = ops.nchw_to_nhwc(gy.to(torch.float32).contiguous(), y.t.dtype, c_pad=y.t.shape[3])
        gfeats, grads = run.backward([])
        ctx.run = None
        return (None, None, None, gfeats[0]) + tuple(grads.get(n) for n in ctx.names)


def _run_block(mod, x):
    if x.shape[1] % 8:
        raise BtsAmdError("standalone %s: input channels must be a multiple of 8 (got %d)" % (type(mod).__name__, x.shape[1]))
    names = tuple(n for n, _ in mod.named_parameters())
    params = tuple(p for _, p in mod.named_parameters())
    record = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    src = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
    return _BlockFn.apply(mod, names, record, src.contiguous(), *params)


class atrous_conv(nn.Sequential):
    """bts.py:51-66.  Child names: atrous_conv.{first_bn, aconv_sequence.{0..4}}.  Inside ``bts`` the fused decoder
    executes it; called on its own it runs the same kernels through ``_BlockFn``."""

    def __init__(self, in_channels, out_channels, dilation, apply_bn_first=True):
        super().__init__()
        self.atrous_conv = torch.nn.Sequential()
        if apply_bn_first:
            self.atrous_conv.add_module("first_bn", nn.BatchNorm2d(in_channels, momentum=0.01, affine=True,
                                                                   track_running_stats=True, eps=1.1e-5))
        self.atrous_conv.add_module("aconv_sequence", nn.Sequential(
            nn.ReLU(),
            nn.Conv2d(in_channels, out_channels * 2, kernel_size=1, stride=1, padding=0, bias=False),
            nn.BatchNorm2d(out_channels * 2, momentum=0.01, affine=True, track_running_stats=True),
            nn.ReLU(),
            nn.Conv2d(out_channels * 2, out_channels, kernel_size=3, stride=1, padding=(dilation, dilation),
                      dilation=dilation, bias=False)))

        self._dil, self._first, self._cout = dilation, apply_bn_first, out_channels
        p = "atrous_conv.aconv_sequence"
        self._plan = _BlockPlan({p + ".1": ConvLayer(p + ".1", out_channels * 2, [in_channels], 1),
                                 p + ".4": ConvLayer(p + ".4", out_channels, [out_channels * 2], 9, dilation)})
        self.compute_dtype = torch.float32

    def _schedule(self, run, a):
        p = "atrous_conv.aconv_sequence"
        n = run.bn(a, "atrous_conv.first_bn", 1.1e-5, relu=True) if self._first else run.relu(a)      # bts.py:54-57
        c = run.conv(p + ".1", [n], ACT_NONE)
        c = run.bn(c, p + ".2", 1e-5, relu=True)                                                    # bts.py:60-61
        return run.conv(p + ".4", [c], ACT_NONE)

    def forward(self, x):
        return _run_block(self, x)


class upconv(nn.Module):
    """bts.py:69-80 (container)."""

    def __init__(self, in_channels, out_channels, ratio=2):
        super().__init__()
        self.elu = nn.ELU()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.ratio = ratio

        if ratio != 2:
            raise BtsAmdError("upconv: only ratio 2 (the only one bts.py uses) has a kernel")
        self._cout = out_channels
        self._plan = _BlockPlan({"conv": ConvLayer("conv", out_channels, [in_channels], 9, 1, True)})
        self.compute_dtype = torch.float32

    def _schedule(self, run, a):
        return run.conv("conv", [a], ACT_ELU)               # nearest x2 folded into 4 sub-pixel phases (bts.py:76-80)

    def forward(self, x):
        return _run_block(self, x)


class reduction_1x1(nn.Sequential):
    """bts.py:83-122 (container); child names inter_<in>_<out> / plane_params / final."""

    def __init__(self, num_in_filters, num_out_filters, max_depth, is_final=False):
        super().__init__()
        self.max_depth = max_depth
        self.is_final = is_final
        self.sigmoid = nn.Sigmoid()
        self.reduc = torch.nn.Sequential()
        for child, cin, cout in reduction_specs(num_in_filters, num_out_filters, is_final):
            conv = nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False)
            if child == "plane_params":
                self.reduc.add_module(child, conv)
            elif child == "final":
                self.reduc.add_module(child, torch.nn.Sequential(conv, nn.Sigmoid()))
            else:
                self.reduc.add_module(child, torch.nn.Sequential(conv, nn.ELU()))

        layers, keys = {}, []
        for child, cin, cout in reduction_specs(num_in_filters, num_out_filters, is_final):
            key = "reduc.%s" % (child + (".0" if child != "plane_params" else ""))
            layers[key] = ConvLayer(key, cout, [cin], 1)
            keys.append(key)
        self._chain = "reduc1x1" if is_final else "reduc"           # DecoderRun.chain ends with the sigmoid map for reduc1x1
        self._plan = _BlockPlan(layers, {self._chain: keys})
        self._cout = 1 if is_final else 4
        self.compute_dtype = torch.float32

    def _schedule(self, run, a):
        raw = run.chain(self._chain, a)                              # 1x1 + ELU chain on the HIP kernels (bts.py:110)
        return raw if self.is_final else run.plane(raw)              # sigmoid map (bts.py:93-96) / plane parameters (:112-120)

    def forward(self, net):
        return _run_block(self, net)                                 # [B,1,h,w] sigmoid map, or [B,4,h,w] (n1, n2, n3, dist)


class _LpgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane_eq, k):
        require_gpu(plane_eq)
        eq = plane_eq.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()   # NCHW -> the op's [B,h,w,4]
        ctx.save_for_backward(eq)
        ctx.k = k
        return ops.lpg_fwd(eq, k)

    @staticmethod
    def backward(ctx, g):
        (eq,) = ctx.saved_tensors
        geq = ops.lpg_bwd(g.to(torch.float32).contiguous(), eq, ctx.k)
        return geq.permute(0, 3, 1, 2), None


class local_planar_guidance(nn.Module):
    """bts.py:124-146: plane_eq [B,4,h,w] -> depth [B,k*h,k*w] via the LPG kernel (focal ignored)."""

    def __init__(self, upratio):
        super().__init__()
        self.upratio = float(upratio)
        self.u = torch.arange(int(upratio)).reshape([1, 1, int(upratio)]).float()
        self.v = torch.arange(int(upratio)).reshape([1, int(upratio), 1]).float()

    def forward(self, plane_eq, focal):
        return _LpgFn.apply(plane_eq, int(self.upratio))


# ---------------------------------------------------------------------------------------------
# decoder
# ---------------------------------------------------------------------------------------------
class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, focal, n_feat, names, record, *tensors):
        feats, params = tensors[:n_feat], tensors[n_feat:]
        P = dict(zip(names, (p.detach() for p in params)))
        for k, b in mod.named_buffers():
            P[k] = b
        bn_training = {k: m.training for k, m in mod.named_modules() if isinstance(m, nn.BatchNorm2d)}
        run = DecoderRun(mod._plan, P, bn_training, mod.params.max_depth, mod.params.dataset, mod.compute_dtype, record)
        outs = run.forward([f.detach() for f in feats], focal)
        ctx.run, ctx.names, ctx.n_feat = run, names, n_feat
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.run is None:
            raise BtsAmdError("the bts decoder keeps one backward tape per forward pass: a second backward() through the "
                              "same forward (retain_graph=True) is not supported -- run the forward again")
        gfeats, grads = ctx.run.backward(gouts)
        ctx.run = None
        gp = [grads.get(n) for n in ctx.names]
        return (None, None, None, None, None) + tuple(gfeats) + tuple(gp)


class bts(nn.Module):
    """Decoder (bts.py:148-266): same submodule names/order; forward runs the HIP executor."""

    def __init__(self, params, feat_out_channels, num_features=512):
        super().__init__()
        self.params = params
        nf, f = num_features, feat_out_channels
        md = self.params.max_depth

        def bn(c):
            return nn.BatchNorm2d(c, momentum=0.01, affine=True, eps=1.1e-5)

        def conv_elu(cin, cout):
            return torch.nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.ELU())

        self.upconv5 = upconv(f[4], nf)
        self.bn5 = bn(nf)
        self.conv5 = conv_elu(nf + f[3], nf)
        self.upconv4 = upconv(nf, nf // 2)
        self.bn4 = bn(nf // 2)
        self.conv4 = conv_elu(nf // 2 + f[2], nf // 2)
        self.bn4_2 = bn(nf // 2)
        self.daspp_3 = atrous_conv(nf // 2, nf // 4, 3, apply_bn_first=False)
        self.daspp_6 = atrous_conv(nf // 2 + nf // 4 + f[2], nf // 4, 6)
        self.daspp_12 = atrous_conv(nf + f[2], nf // 4, 12)
        self.daspp_18 = atrous_conv(nf + nf // 4 + f[2], nf // 4, 18)
        self.daspp_24 = atrous_conv(nf + nf // 2 + f[2], nf // 4, 24)
        self.daspp_conv = conv_elu(nf + nf // 2 + nf // 4, nf // 4)
        self.reduc8x8 = reduction_1x1(nf // 4, nf // 4, md)
        self.lpg8x8 = local_planar_guidance(8)
        self.upconv3 = upconv(nf // 4, nf // 4)
        self.bn3 = bn(nf // 4)
        self.conv3 = conv_elu(nf // 4 + f[1] + 1, nf // 4)
        self.reduc4x4 = reduction_1x1(nf // 4, nf // 8, md)
        self.lpg4x4 = local_planar_guidance(4)
        self.upconv2 = upconv(nf // 4, nf // 8)
        self.bn2 = bn(nf // 8)
        self.conv2 = conv_elu(nf // 8 + f[0] + 1, nf // 8)
        self.reduc2x2 = reduction_1x1(nf // 8, nf // 16, md)
        self.lpg2x2 = local_planar_guidance(2)
        self.upconv1 = upconv(nf // 8, nf // 16)
        self.reduc1x1 = reduction_1x1(nf // 16, nf // 32, md, is_final=True)
        self.conv1 = conv_elu(nf // 16 + 4, nf // 16)
        self.get_depth = torch.nn.Sequential(nn.Conv2d(nf // 16, 1, 3, 1, 1, bias=False), nn.Sigmoid())

        self._plan = DecoderPlan(f, nf)
        self._param_names = tuple(n for n, _ in self.named_parameters())
        self._param_getters = tuple(operator.attrgetter(n) for n in self._param_names)
        # activation dtype of the decoder kernels: f32 (parity) or bf16 (throughput); not a parameter
        self.compute_dtype = getattr(params, "decoder_dtype", torch.float32)
        if isinstance(self.compute_dtype, str):
            self.compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32,
                                  "float32": torch.float32}[self.compute_dtype]

    def forward(self, features, focal):
        feats = list(features[:5])
        require_gpu(feats[0])
        # parameters by cached dotted name: nn.DataParallel replicas (bts_main.py:357, bts_test.py:90) carry them as
        # plain tensor attributes and expose nothing through named_parameters()
        names = self._param_names
        params = tuple(g(self) for g in self._param_getters)
        # record the backward tape only when autograd will ask for it; the no-grad path (bts_test.py:118-119,
        # online_eval) runs the fused inference LPG heads and keeps no activations
        record = torch.is_grad_enabled() and (any(f.requires_grad for f in feats) or any(p.requires_grad for p in params))
        return _DecoderFn.apply(self, focal, len(feats), names, record, *feats, *params)


# ---------------------------------------------------------------------------------------------
# encoder (bts.py:268-320): stock PyTorch-ROCm backbone, by design
# ---------------------------------------------------------------------------------------------
def _tv_models():
    try:
        import torchvision.models as models   # the reference's source of backbones (bts.py:272)
        return models, True
    except ImportError:
        from . import tv_models                 # same architectures / names, random init
        return tv_models, False


class encoder(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        models, real = _tv_models()
        pre = dict(pretrained=True) if real and getattr(params, "pretrained", True) else {}
        e = params.encoder
        if e in ("densenet121_bts", "densenet161_bts"):
            self.base_model = getattr(models, e[:-4])(**pre).features
            self.feat_names = ["relu0", "pool0", "transition1", "transition2", "norm5"]
            self.feat_out_channels = [64, 64, 128, 256, 1024] if e == "densenet121_bts" else [96, 96, 192, 384, 2208]
        elif e in ("resnet50_bts", "resnet101_bts", "resnext50_bts", "resnext101_bts"):
            ctor = {"resnet50_bts": "resnet50", "resnet101_bts": "resnet101", "resnext50_bts": "resnext50_32x4d",
                    "resnext101_bts": "resnext101_32x8d"}[e]
            self.base_model = getattr(models, ctor)(**pre)
            self.feat_names = ["relu", "layer1", "layer2", "layer3", "layer4"]
            self.feat_out_channels = [64, 256, 512, 1024, 2048]
        elif e == "mobilenetv2_bts":
            self.base_model = models.mobilenet_v2(**pre).features
            self.feat_inds = [2, 4, 7, 11, 19]
            self.feat_out_channels = [16, 24, 32, 64, 1280]
            self.feat_names = []
        else:
            print("Not supported encoder: {}".format(params.encoder))    # bts.py:303

    def _is_tap(self, index, name):
        """Which backbone children feed the decoder (bts.py:311-319): by position for MobileNetV2,
        by substring of the child name otherwise."""
        if self.params.encoder == "mobilenetv2_bts":
            return index in self.feat_inds
        return any(tag in name for tag in self.feat_names)

    def forward(self, x):
        taps = []
        stages = [(n, m) for n, m in self.base_model._modules.items() if "fc" not in n and "avgpool" not in n]
        for index, (name, stage) in enumerate(stages, start=1):
            x = stage(x)
            if self._is_tap(index, name):
                taps.append(x)
        return taps


class BtsModel(nn.Module):
    """bts.py:323-331."""

    def __init__(self, params):
        super().__init__()
        self.encoder = encoder(params)
        self.decoder = bts(params, self.encoder.feat_out_channels, params.bts_size)

    def forward(self, x, focal):
        skip_feat = self.encoder(x)
        return self.decoder(skip_feat, focal)
