"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the BTS decoder / LPG / silog hot path.

A functional restatement (plain PyTorch CPU ops, fp32 or fp64) of the algorithm
in the reference's ``pytorch/bts.py``.  It is written from the reference's
*behaviour* (each function cites the lines it follows) and takes the decoder
parameters as a flat ``{state_dict_key: tensor}`` mapping using the reference's
key names, so the same weights can be fed to the
reference module, to this oracle and to the HIP product path.

Pinning: the reference ships no tests or golden vectors, so this
oracle is pinned against outputs of the reference module itself, generated in
the build container by ``tools/make_golden.py`` and committed under
``tests/golden/`` (checked by ``tests/test_oracle_golden.py``), and, where
``/root/reference`` is present, against the live reference
(``tests/test_oracle_golden.py::test_oracle_vs_live_reference_c1_plumbing``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``bts_amd/`` does.
"""
import math

import torch
import torch.nn.functional as F

KITTI_FOCAL_REF = 715.0873  # bts.py:264


# --------------------------------------------------------------------------
# local planar guidance -- bts.py:124-146
# --------------------------------------------------------------------------
def lpg(plane_eq, k):
    """plane_eq [B,4,h,w] -> depth [B,k*h,k*w].

    bts.py:133-134 replicate every coarse cell k x k; bts.py:140-144 build
    u = ((col mod k) - (k-1)/2)/k and v likewise from the row; bts.py:146
    depth = n4 / (n1*u + n2*v + n3).  ``focal`` is accepted and ignored by the
    reference (bts.py:132) so it is not a parameter here.
    """
    B, _, h, w = plane_eq.shape
    dt = plane_eq.dtype
    off = (torch.arange(k, dtype=dt, device=plane_eq.device) - (k - 1) * 0.5) / k          # [k]
    u = off.repeat(w).view(1, 1, w * k)                            # varies with column
    v = off.repeat(h).view(1, h * k, 1)                            # varies with row
    e = plane_eq.repeat_interleave(k, 2).repeat_interleave(k, 3)
    return e[:, 3] / (e[:, 0] * u + e[:, 1] * v + e[:, 2])


def plane_from_raw(raw, max_depth):
    """bts.py:112-120: raw [B,3,h,w] (last 1x1 conv, no activation) -> [B,4,h,w]."""
    theta = torch.sigmoid(raw[:, 0]) * math.pi / 3
    phi = torch.sigmoid(raw[:, 1]) * math.pi * 2
    dist = torch.sigmoid(raw[:, 2]) * max_depth
    n1 = torch.sin(theta) * torch.cos(phi)
    n2 = torch.sin(theta) * torch.sin(phi)
    n3 = torch.cos(theta)
    return torch.stack([n1, n2, n3, dist], 1)


def normalize_plane(eq):
    """bts.py:223-226 (and 237-240, 251-254): L2-normalise the normal, keep n4."""
    n = eq[:, :3]
    n = n / n.norm(dim=1, keepdim=True).clamp_min(1e-12)           # F.normalize semantics
    return torch.cat([n, eq[:, 3:4]], 1)


def reduction_chain(x, P, prefix, max_depth, is_final):
    """reduction_1x1 (bts.py:83-122): 1x1 conv + ELU chain, then plane params / final."""
    keys = [k for k in P if k.startswith(prefix + ".reduc.")]
    for k in keys:
        w = P[k]
        x = F.conv2d(x, w)
        if ".inter_" in k:
            x = F.elu(x)                                           # bts.py:101-105
    if is_final:
        return torch.sigmoid(x)                                    # bts.py:93-96
    return plane_from_raw(x, max_depth)                            # bts.py:112-120


# --------------------------------------------------------------------------
# decoder building blocks
# --------------------------------------------------------------------------
class BNState:
    """Collects running-stat updates (momentum 0.01, unbiased var) like nn.BatchNorm2d."""

    def __init__(self, training):
        self.training = training
        self.updates = {}


def _bn(x, P, prefix, eps, st, momentum=0.01):
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if st.training:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        st.updates[prefix + ".running_mean"] = (1 - momentum) * P[prefix + ".running_mean"] + momentum * mean.detach()
        st.updates[prefix + ".running_var"] = (1 - momentum) * P[prefix + ".running_var"] + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    s = w / torch.sqrt(var + eps)
    return x * s.view(1, -1, 1, 1) + (b - mean * s).view(1, -1, 1, 1)


def _upconv(x, w):
    """upconv (bts.py:69-80): nearest x2 -> 3x3 conv (pad 1, no bias) -> ELU."""
    x = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    return F.elu(F.conv2d(x, w, padding=1))


def _atrous(x, P, prefix, d, st, first_bn):
    """atrous_conv (bts.py:51-66): [BN eps 1.1e-5] ReLU 1x1 BN(eps 1e-5) ReLU 3x3 dil d."""
    if first_bn:
        x = _bn(x, P, prefix + ".atrous_conv.first_bn", 1.1e-5, st)
    x = F.conv2d(F.relu(x), P[prefix + ".atrous_conv.aconv_sequence.1.weight"])
    x = _bn(x, P, prefix + ".atrous_conv.aconv_sequence.2", 1e-5, st)
    return F.conv2d(F.relu(x), P[prefix + ".atrous_conv.aconv_sequence.4.weight"], padding=d, dilation=d)


def decoder_forward(P, features, focal, max_depth, dataset, training=True):
    """bts.forward (bts.py:196-266).  Returns (outputs5, bn_running_updates)."""
    st = BNState(training)
    skip0, skip1, skip2, skip3 = features[0], features[1], features[2], features[3]
    dense = F.relu(features[4])                                              # :198
    up5 = _bn(_upconv(dense, P["upconv5.conv.weight"]), P, "bn5", 1.1e-5, st)  # :199-200
    cat5 = torch.cat([up5, skip3], 1)
    i5 = F.elu(F.conv2d(cat5, P["conv5.0.weight"], padding=1))               # :202
    up4 = _bn(_upconv(i5, P["upconv4.conv.weight"]), P, "bn4", 1.1e-5, st)   # :204-205
    cat4 = torch.cat([up4, skip2], 1)
    i4 = F.elu(F.conv2d(cat4, P["conv4.0.weight"], padding=1))               # :207
    i4 = _bn(i4, P, "bn4_2", 1.1e-5, st)                                     # :208
    d3 = _atrous(i4, P, "daspp_3", 3, st, False)                             # :210
    c = torch.cat([cat4, d3], 1)
    d6 = _atrous(c, P, "daspp_6", 6, st, True)                               # :212
    c = torch.cat([c, d6], 1)
    d12 = _atrous(c, P, "daspp_12", 12, st, True)                            # :214
    c = torch.cat([c, d12], 1)
    d18 = _atrous(c, P, "daspp_18", 18, st, True)                            # :216
    c = torch.cat([c, d18], 1)
    d24 = _atrous(c, P, "daspp_24", 24, st, True)                            # :218
    dfeat = F.elu(F.conv2d(torch.cat([i4, d3, d6, d12, d18, d24], 1),
                           P["daspp_conv.0.weight"], padding=1))             # :219-220

    eq8 = normalize_plane(reduction_chain(dfeat, P, "reduc8x8", max_depth, False))  # :222-226
    d8 = lpg(eq8, 8).unsqueeze(1) / max_depth                                # :227-228
    d8_ds = d8[:, :, ::4, ::4]                                               # :229 nearest x0.25

    up3 = _bn(_upconv(dfeat, P["upconv3.conv.weight"]), P, "bn3", 1.1e-5, st)   # :231-232
    i3 = F.elu(F.conv2d(torch.cat([up3, skip1, d8_ds], 1), P["conv3.0.weight"], padding=1))
    eq4 = normalize_plane(reduction_chain(i3, P, "reduc4x4", max_depth, False))  # :236-240
    d4 = lpg(eq4, 4).unsqueeze(1) / max_depth
    d4_ds = d4[:, :, ::2, ::2]                                               # :243 nearest x0.5

    up2 = _bn(_upconv(i3, P["upconv2.conv.weight"]), P, "bn2", 1.1e-5, st)   # :245-246
    i2 = F.elu(F.conv2d(torch.cat([up2, skip0, d4_ds], 1), P["conv2.0.weight"], padding=1))
    eq2 = normalize_plane(reduction_chain(i2, P, "reduc2x2", max_depth, False))  # :250-254
    d2 = lpg(eq2, 2).unsqueeze(1) / max_depth

    up1 = _upconv(i2, P["upconv1.conv.weight"])                              # :258 (no BN)
    r1 = reduction_chain(up1, P, "reduc1x1", max_depth, True)                # :259
    i1 = F.elu(F.conv2d(torch.cat([up1, r1, d2, d4, d8], 1), P["conv1.0.weight"], padding=1))
    depth = max_depth * torch.sigmoid(F.conv2d(i1, P["get_depth.0.weight"], padding=1))  # :262
    if dataset == "kitti":
        depth = depth * focal.view(-1, 1, 1, 1).to(depth.dtype) / KITTI_FOCAL_REF      # :263-264
    return (d8, d4, d2, r1, depth), st.updates


# --------------------------------------------------------------------------
# silog loss -- bts.py:41-48
# --------------------------------------------------------------------------
def silog(depth_est, depth_gt, mask, variance_focus):
    """10 * sqrt(mean(d^2) - vf * mean(d)^2), d = log(est) - log(gt) over mask."""
    d = torch.log(depth_est[mask]) - torch.log(depth_gt[mask])
    return torch.sqrt((d * d).mean() - variance_focus * d.mean() ** 2) * 10.0


# --------------------------------------------------------------------------
# synthetic inputs shared by tests / bench (SURVEY.md section 8c "oracle recipe")
# --------------------------------------------------------------------------
KITTI_FOCALS = (721.5377, 718.856, 707.0912, 718.3351, 707.0493)   # train_test_inputs/eigen_*.txt col 3
NYU_FOCAL = 518.8579                                               # train_test_inputs/nyudepthv2_*.txt col 3


def synth_focal(batch, dataset):
    vals = KITTI_FOCALS if dataset == "kitti" else (NYU_FOCAL,)
    return torch.tensor([vals[i % len(vals)] for i in range(batch)], dtype=torch.float64)


def synth_depth_gt(batch, H, W, dataset, gen):
    max_depth = 80.0 if dataset == "kitti" else 10.0
    lo = 1.5 if dataset == "kitti" else 0.5
    gt = torch.rand(batch, 1, H, W, generator=gen) * (max_depth - 0.5 - lo) + lo
    hole = torch.rand(batch, 1, H, W, generator=gen) < 0.3
    return gt.masked_fill(hole, 0.0)


# --------------------------------------------------------------------------
# decoder parameter factory (names / shapes / order of bts.__init__, bts.py:149-194;
# init = weights_init_xavier bts.py:33-38 for convs, nn.BatchNorm2d defaults)
# --------------------------------------------------------------------------
def reduction_specs(c_in, c_out, is_final):
    """(key-suffix, cin, cout) list of the 1x1 convs built by reduction_1x1.__init__ (bts.py:91-108)."""
    out = []
    while c_out >= 4:
        if c_out < 8:
            out.append(("final.0" if is_final else "plane_params", c_in, 1 if is_final else 3))
            break
        out.append(("inter_%d_%d.0" % (c_in, c_out), c_in, c_out))
        c_in, c_out = c_out, c_out // 2
    return out


def decoder_param_specs(feat, nf):
    """Ordered [(key, shape, kind)] for the decoder state dict; kind in conv|bn_w|bn_b|bn_rm|bn_rv|bn_nbt."""
    S = []

    def conv(name, co, ci, k):
        S.append((name + ".weight", (co, ci, k, k), "conv"))

    def bn(name, c):
        S.extend([(name + ".weight", (c,), "bn_w"), (name + ".bias", (c,), "bn_b"),
                  (name + ".running_mean", (c,), "bn_rm"), (name + ".running_var", (c,), "bn_rv"),
                  (name + ".num_batches_tracked", (), "bn_nbt")])

    def atrous(name, ci, co, first_bn):
        if first_bn:
            bn(name + ".atrous_conv.first_bn", ci)
        conv(name + ".atrous_conv.aconv_sequence.1", co * 2, ci, 1)
        bn(name + ".atrous_conv.aconv_sequence.2", co * 2)
        conv(name + ".atrous_conv.aconv_sequence.4", co, co * 2, 3)

    def reduc(name, ci, co, final=False):
        for suffix, a, b in reduction_specs(ci, co, final):
            conv(name + ".reduc." + suffix, b, a, 1)

    conv("upconv5.conv", nf, feat[4], 3); bn("bn5", nf)
    conv("conv5.0", nf, nf + feat[3], 3)
    conv("upconv4.conv", nf // 2, nf, 3); bn("bn4", nf // 2)
    conv("conv4.0", nf // 2, nf // 2 + feat[2], 3); bn("bn4_2", nf // 2)
    atrous("daspp_3", nf // 2, nf // 4, False)
    atrous("daspp_6", nf // 2 + nf // 4 + feat[2], nf // 4, True)
    atrous("daspp_12", nf + feat[2], nf // 4, True)
    atrous("daspp_18", nf + nf // 4 + feat[2], nf // 4, True)
    atrous("daspp_24", nf + nf // 2 + feat[2], nf // 4, True)
    conv("daspp_conv.0", nf // 4, nf + nf // 2 + nf // 4, 3)
    reduc("reduc8x8", nf // 4, nf // 4)
    conv("upconv3.conv", nf // 4, nf // 4, 3); bn("bn3", nf // 4)
    conv("conv3.0", nf // 4, nf // 4 + feat[1] + 1, 3)
    reduc("reduc4x4", nf // 4, nf // 8)
    conv("upconv2.conv", nf // 8, nf // 4, 3); bn("bn2", nf // 8)
    conv("conv2.0", nf // 8, nf // 8 + feat[0] + 1, 3)
    reduc("reduc2x2", nf // 8, nf // 16)
    conv("upconv1.conv", nf // 16, nf // 8, 3)
    reduc("reduc1x1", nf // 16, nf // 32, True)
    conv("conv1.0", nf // 16, nf // 16 + 4, 3)
    conv("get_depth.0", 1, nf // 16, 3)
    return S


def make_decoder_params(feat, nf, gen, dtype=torch.float32, randomize_bn=False):
    """Seeded decoder parameters.  randomize_bn perturbs BN affine/running stats so that
    tests exercise non-trivial gamma/beta/mean/var (default nn.BatchNorm2d init is 1/0/0/1)."""
    from collections import OrderedDict
    P = OrderedDict()
    for key, shape, kind in decoder_param_specs(feat, nf):
        if kind == "conv":
            co, ci, k, _ = shape
            bound = math.sqrt(6.0 / ((ci + co) * k * k))
            P[key] = ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        elif kind == "bn_nbt":
            P[key] = torch.zeros((), dtype=torch.long)
        elif not randomize_bn:
            P[key] = (torch.ones if kind in ("bn_w", "bn_rv") else torch.zeros)(shape, dtype=dtype)
        else:
            r = torch.rand(shape, generator=gen, dtype=torch.float64)
            P[key] = {"bn_w": 0.5 + r, "bn_b": r - 0.5, "bn_rm": 0.2 * (r - 0.5), "bn_rv": 0.5 + r}[kind].to(dtype)
    return P


def make_features(feat, batch, H, W, gen, dtype=torch.float32):
    """Encoder feature maps at H/2 .. H/32 (bts.py:276-296 channel lists), ~N(0,1)."""
    return [torch.randn(batch, c, H >> (i + 1), W >> (i + 1), generator=gen, dtype=torch.float64).to(dtype)
            for i, c in enumerate(feat)]
