"""GPU tests at BASELINE.json's full sizes.  The CPU oracle is too slow there, so these use
size-independent properties of the path (homogeneity / linearity of LPG, Euler identities of its gradient,
determinism of the forward pass, linearity of the convolution) plus the oracle's own formulas evaluated with
torch ops ON THE DEVICE as the checker (never as the product path)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("k", [8, 4, 2])
@pytest.mark.parametrize("shape", [(8, 352, 1216), (32, 704, 1216)], ids=["train_c3", "infer_c5"])
def test_lpg_full_size_properties(k, shape):
    from bts_amd import ops
    B, H, W = shape
    h, w = H // k, W // k
    gen = torch.Generator(device=DEV).manual_seed(k)
    raw = torch.randn(B, 3, h, w, device=DEV, generator=gen)
    eq = O.normalize_plane(O.plane_from_raw(raw, 80.0)).permute(0, 2, 3, 1).contiguous()
    d1 = ops.lpg_fwd(eq, k)
    # (1) exact oracle formula evaluated on device: bit-identical
    ref = O.lpg(eq.permute(0, 3, 1, 2).cpu()[:1], k)
    assert torch.equal(d1[:1].cpu(), ref)
    # (2) degree-1 homogeneity in n4: scaling by a power of two is exact
    eq2 = eq.clone()
    eq2[..., 3] *= 4.0
    assert torch.equal(ops.lpg_fwd(eq2, k), d1 * 4.0)
    # (3) cell constancy of the plane: depth at (u, v) and (-u, -v) of a fronto-parallel plane (n1=n2=0) are equal
    eq3 = eq.clone()
    eq3[..., 0] = 0
    eq3[..., 1] = 0
    d3 = ops.lpg_fwd(eq3, k).view(B, h, k, w, k)
    assert torch.equal(d3, d3.flip(2).flip(4))
    # (4) Euler identities of the gradient: <g, d> = sum g4*n4 = -sum (g1 n1 + g2 n2 + g3 n3)
    g = torch.randn(B, H, W, device=DEV, generator=gen)
    geq = ops.lpg_bwd(g, eq, k)
    lhs = (g.double() * d1.double()).sum()
    e4 = (geq[..., 3].double() * eq[..., 3].double()).sum()
    e123 = (geq[..., :3].double() * eq[..., :3].double()).sum()
    assert abs((e4 - lhs) / lhs) < 1e-4 and abs((e123 + lhs) / lhs) < 1e-4
    # (5) fused head == plane math + LPG op, and its depth_div
    rawn = torch.zeros(B, h, w, 4, device=DEV)
    rawn[..., :3] = raw.permute(0, 2, 3, 1)
    dh = ops.lpg_head_fwd(rawn, k, 80.0)
    # steep planes make n1*u + n2*v + n3 pass through zero (|u|,|v| up to 7/16, theta up to pi/3): there the depth is
    # ill-conditioned w.r.t. 1-ulp differences of sin/cos between this kernel and torch; compare where it is not
    ok = (d1 > 0) & (d1 < 160.0)
    assert ok.float().mean().item() > 0.9
    assert rel(dh[ok], d1[ok] / 80.0) < 1e-4


def test_silog_full_size_vs_device_formula():
    from bts_amd.model import silog_loss
    B, H, W = 8, 352, 1216
    gen = torch.Generator().manual_seed(3)
    gt = O.synth_depth_gt(B, H, W, "kitti", gen).to(DEV)
    est = (torch.rand(B, 1, H, W, generator=gen) * 79 + 0.5).to(DEV).requires_grad_(True)
    mask = gt > 1.0
    loss = silog_loss(0.85)(est, gt, mask)
    est2 = est.detach().double().requires_grad_(True)
    ref = O.silog(est2, gt.double(), mask, 0.85)          # oracle formula, f64, on device
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-5
    loss.backward()
    ref.backward()
    assert rel(est.grad, est2.grad) < 1e-4
    # scale invariance of the variance term: vf = 1 makes the loss invariant to est -> c*est
    a = silog_loss(1.0)(est.detach(), gt, mask).item()
    b = silog_loss(1.0)(est.detach() * 3.0, gt, mask).item()
    assert abs(a - b) / a < 1e-3


def test_decoder_full_size_vs_device_oracle_f32():
    """DenseNet161 widths, 352x1216, batch 2, f32: HIP decoder vs the oracle's formulas run with torch ops on the device."""
    from bts_amd.model import bts, silog_loss
    feat, nf, B, H, W = [96, 96, 192, 384, 2208], 512, 2, 352, 1216
    gen = torch.Generator().manual_seed(77)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    gt = O.synth_depth_gt(B, H, W, "kitti", gen).to(DEV)
    Pd = {k: v.to(DEV) for k, v in P.items()}
    prev = torch.backends.cudnn.allow_tf32
    with torch.no_grad():
        ref, upd = O.decoder_forward(Pd, [f.to(DEV) for f in feats], focal.to(DEV), 80.0, "kitti", True)
    dec = bts(NS(max_depth=80.0, dataset="kitti", encoder="densenet161_bts", bts_size=nf), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).train()
    fs = [f.to(DEV).requires_grad_(True) for f in feats]
    outs = dec(fs, focal.to(DEV))
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert rel(o, r) < 1e-4, i
    loss = silog_loss(0.85)(outs[4], gt, gt > 1.0)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.parameters())
    assert all(f.grad is not None and torch.isfinite(f.grad).all() for f in fs)
    for k_, v in upd.items():
        assert rel(dec.state_dict()[k_], v) < 1e-4, k_
    # determinism of the forward pass (no atomics on the forward path): eval twice -> bit-identical
    dec.eval()
    with torch.no_grad():
        a = dec([f.detach() for f in fs], focal.to(DEV))
        b = dec([f.detach() for f in fs], focal.to(DEV))
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_conv_linearity_full_size_bf16():
    """conv2-shaped layer (64 <- [64, 96, 1] @ 176x608, batch 8) in bf16: linear in the weights."""
    from bts_amd._lib import ACT_NONE
    from bts_amd.conv import ConvLayer
    N, H, W = 8, 176, 608
    L = ConvLayer("conv2", 64, [64, 96, 1], 9)
    gen = torch.Generator(device=DEV).manual_seed(5)
    segs = [torch.randn(N, H, W, c, device=DEV, generator=gen).to(torch.bfloat16) for c in (64, 96, 8)]
    segs[2][..., 1:] = 0
    w1 = (torch.randn(64, 161, 3, 3, device=DEV, generator=gen) * 0.03).to(torch.bfloat16).float()
    w2 = (torch.randn(64, 161, 3, 3, device=DEV, generator=gen) * 0.03).to(torch.bfloat16).float()
    outs = []
    for w in (w1, w2, w1 + w2):
        o = torch.empty(N, H, W, 64, dtype=torch.float32, device=DEV)
        L.forward(segs, L.pack_fwd(w, torch.bfloat16), o, ACT_NONE)
        outs.append(o)
    # w1+w2 is re-rounded to bf16 when packed: allow bf16 weight rounding, nothing more
    assert rel(outs[0] + outs[1], outs[2]) < 2e-2
    assert torch.isfinite(outs[2]).all()


def l2rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# BASELINE.json configs[2] per-GPU share (8 x 352x1216, kitti), configs[1] (16 x 416x544, nyu) -- DenseNet161 widths -- and
# configs[3] per-GPU share (8 x 352x1216, kitti, f32) with the ResNet/ResNeXt-family widths of pytorch/bts.py:280-296
# (feat_out_channels = [64, 256, 512, 1024, 2048]: conv3 sees 128+256+1 channels, conv2 64+64+1, upconv5 2048 -> 512).
# Bounds (L2-relative unless noted), stated here and in DESIGN.md section 2:
#   f32 : outputs 1e-4 max-norm (north_star), loss 1e-5, every parameter / feature gradient 1e-3
#   bf16: (activations + packed weights rounded to bf16, f32 accumulate, fused LPG chains) outputs 1e-2, loss 2e-3,
#         every parameter / feature gradient 3e-2 -- a throughput configuration, bounded per tensor, not a parity claim
DN161, RESNEXT = [96, 96, 192, 384, 2208], [64, 256, 512, 1024, 2048]
BENCH_CONFIGS = {"c3": (8, 352, 1216, "kitti", 80.0, DN161), "c2": (16, 416, 544, "nyu", 10.0, DN161),
                 "c4": (8, 352, 1216, "kitti", 80.0, RESNEXT)}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cfg", ["c3", "c2", "c4"])
def test_decoder_parity_at_bench_config(cfg, dt):
    """The benchmarked configuration itself: full batch, full resolution, the encoder family's widths, the dtype and the fused
    LPG-chain kernels bench.py runs -- five outputs, loss, EVERY parameter gradient and EVERY feature gradient against the
    oracle's formulas (bts.py:196-266, 41-48) evaluated in f32 with torch ops + autograd on the device."""
    import json
    import os

    from bts_amd import profiler
    from bts_amd.model import bts, silog_loss
    B, H, W, ds, md, feat = BENCH_CONFIGS[cfg]
    nf = 512
    gen = torch.Generator().manual_seed(2024)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, ds)
    gt = O.synth_depth_gt(B, H, W, ds, gen).to(DEV)
    mask = gt > (1.0 if ds == "kitti" else 0.1)

    def objective(outs, loss):
        return loss + sum((o * o).mean() for o in outs[:4])      # every head receives gradient

    # ---- checker: oracle on the device, f32, autograd ----
    Pd = {k: (v.to(DEV).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.to(DEV)) for k, v in P.items()}
    fr = [f.to(DEV).requires_grad_(True) for f in feats]
    # MIOpen off for the checker: ATen's own im2col + rocBLAS f32 convolutions need no per-shape solver search / kernel
    # build on a fresh box (that was most of this test's 90 s) and are plain f32 FMA arithmetic
    with torch.backends.cudnn.flags(enabled=False):
        ref, _ = O.decoder_forward(Pd, fr, focal.to(DEV), md, ds, True)
        loss_ref = O.silog(ref[4], gt, mask, 0.85)
        objective(ref, loss_ref).backward()
    ref = [r.detach() for r in ref]
    gref = {k: v.grad for k, v in Pd.items() if v.dtype.is_floating_point and v.requires_grad}
    gfref = [f.grad for f in fr]
    # ---- product path ----
    dec = bts(NS(max_depth=md, dataset=ds, encoder="densenet161_bts" if feat is DN161 else "resnext101_bts", bts_size=nf, decoder_dtype=dt), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).train()
    fs = [f.to(DEV).requires_grad_(True) for f in feats]
    prof = profiler.enable()
    outs = dec(fs, focal.to(DEV))
    loss = silog_loss(0.85)(outs[4], gt, mask)
    objective(outs, loss).backward()
    names = {r[0] for r in prof.records}
    profiler.disable()
    if dt == torch.bfloat16:        # the kernels bench.py runs: fused chain forward + recompute backward, LDS-DMA convs
        assert any(n.startswith("lpg_head_chain_bwd") for n in names), names
        assert any(n.startswith("conv_igemm_dma<bf16") for n in names), names
    # Gradient bounds.  The only non-smooth ops of the decoder are the ReLUs of the dense ASPP (bts.py:51-66, 198); every
    # other activation is ELU / sigmoid.  At this size ~1e-6 of the 13.7 M ReLU inputs per layer sit within f32 rounding of
    # zero, so two correct f32 implementations take a handful of different masks and the gradients of everything at or
    # upstream of the ASPP differ by ~1e-3 L2 -- measured, not assumed: tools/parity_probe.py against the oracle in f64
    # (profiles/r02_parity_probe_c3.json) puts torch's own f32 evaluation 1.6e-3 from f64 on those tensors and the product
    # 1.7e-3, while everything downstream of the ASPP (ELU only) is <= 1e-6 for both.  Hence 1e-4 where the function is
    # smooth, 5e-3 upstream of the ReLUs, in f32; bf16 adds operand rounding on top.
    out_bound, loss_bound, grad_bound = (1e-4, 1e-5, 1e-4) if dt == torch.float32 else (1e-2, 2e-3, 3e-2)
    # bf16 stores activations with 8 mantissa bits, so ~0.4 % of the ReLU inputs land on the other side of zero than in the
    # f32 evaluation: sqrt(0.004) ~ 6 % of gradient energy differs upstream of the ASPP (measured 0.08-0.10 L2 at both
    # configurations, gpurun r02c); any bf16-activation implementation shares this, it is not operand rounding of a kernel.
    relu_bound = 5e-3 if dt == torch.float32 else 0.15
    smooth = ("daspp_conv", "reduc", "upconv3", "bn3", "conv3", "upconv2", "bn2", "conv2", "upconv1", "conv1", "get_depth",
              "feat0", "feat1")
    rep = {"config": cfg, "dtype": str(dt), "outputs_l2": {}, "outputs_max": {}, "grads_l2": {}}
    for i, (o, r) in enumerate(zip(outs, ref)):
        rep["outputs_l2"]["out%d" % i] = l2rel(o, r)
        rep["outputs_max"]["out%d" % i] = rel(o, r)
    rep["loss"] = abs(loss.item() - loss_ref.item()) / loss_ref.item()
    for n, p in dec.named_parameters():
        assert p.grad is not None, n
        rep["grads_l2"][n] = l2rel(p.grad, gref[n])
    for i, (f, g) in enumerate(zip(fs, gfref)):
        rep["grads_l2"]["feat%d" % i] = l2rel(f.grad, g)
    worst = sorted(rep["grads_l2"].items(), key=lambda kv: -kv[1])[:5]
    print("parity %s %s: outputs(max) %s loss %.2e worst grads %s" % (cfg, rep["dtype"], {k: "%.1e" % v for k, v in rep["outputs_max"].items()},
                                                                      rep["loss"], [(k, "%.1e" % v) for k, v in worst]))
    dump = os.environ.get("BTS_PARITY_DUMP")
    if dump:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, "parity_%s_%s.json" % (cfg, "f32" if dt == torch.float32 else "bf16")), "w") as f:
            json.dump(rep, f, indent=1)
    if dt == torch.float32:
        assert max(rep["outputs_max"].values()) < out_bound, rep["outputs_max"]
    else:
        assert max(rep["outputs_l2"].values()) < out_bound, rep["outputs_l2"]
    assert rep["loss"] < loss_bound
    bad = {k: v for k, v in rep["grads_l2"].items() if not v < (grad_bound if k.startswith(smooth) else relu_bound)}
    assert not bad, bad


def test_decoder_gradients_f64_arbitrated():
    """WHICH side of the f32 gradient comparison above carries the 1-2e-3 behind the dense-ASPP ReLUs?  The oracle's formulas in f64 on
    the device are the arbiter; the product's f32 decoder (fused f32 chains included) and torch's own f32 evaluation of the same
    formulas are both measured against it, tensor by tensor (L2-relative), at the benchmarked resolution and widths (2 of the 8
    images: the f64 convolutions of the full batch would take minutes).  Bar: every gradient of the product is within 1e-4 of the
    f64 result, or no further from it than 2x what torch's f32 evaluation is -- i.e. the product is as close to the truth as an f32
    evaluation of bts.py:196-266 under autograd can be expected to be.  (Until round 5 this argument lived outside the suite:
    tools/parity_probe.py, profiles/r02_parity_probe_c3.json.)"""
    from bts_amd.model import bts, silog_loss
    B, H, W, ds, md, feat = 2, 352, 1216, "kitti", 80.0, DN161
    nf = 512
    gen = torch.Generator().manual_seed(2025)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, ds)
    gt = O.synth_depth_gt(B, H, W, ds, gen).to(DEV)
    mask = gt > 1.0

    def objective(outs, loss):
        return loss + sum((o * o).mean() for o in outs[:4])

    def run_oracle(dt):
        Pd = {k: ((v.to(DEV).to(dt)).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
                  else (v.to(DEV).to(dt) if v.dtype.is_floating_point else v.to(DEV))) for k, v in P.items()}
        fr = [f.to(DEV).to(dt).requires_grad_(True) for f in feats]
        with torch.backends.cudnn.flags(enabled=False):
            ref, _ = O.decoder_forward(Pd, fr, focal.to(DEV).to(dt), md, ds, True)
            loss = O.silog(ref[4], gt.to(dt), mask, 0.85)
            objective(ref, loss).backward()
        g = {k: v.grad.detach() for k, v in Pd.items() if v.dtype.is_floating_point and v.requires_grad}
        for i, f in enumerate(fr):
            g["feat%d" % i] = f.grad.detach()
        return [r.detach() for r in ref], g
    o64, g64 = run_oracle(torch.float64)
    o32, g32 = run_oracle(torch.float32)
    dec = bts(NS(max_depth=md, dataset=ds, encoder="densenet161_bts", bts_size=nf, decoder_dtype=torch.float32), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).train()
    fs = [f.to(DEV).requires_grad_(True) for f in feats]
    outs = dec(fs, focal.to(DEV))
    loss = silog_loss(0.85)(outs[4], gt, mask)
    objective(outs, loss).backward()
    gp = {n: p.grad for n, p in dec.named_parameters()}
    for i, f in enumerate(fs):
        gp["feat%d" % i] = f.grad
    for i in range(5):                       # outputs, element-wise against f64 (north_star's bound)
        r = o64[i].double()
        e = ((outs[i].double() - r).abs() / r.abs().clamp_min(1e-30)).max().item()
        assert e < 1e-4, (i, e)
    worst, bad = (None, 0.0, 0.0), {}
    for k, g in g64.items():
        ep, et = l2rel(gp[k], g), l2rel(g32[k], g)
        if ep > worst[1]:
            worst = (k, ep, et)
        if not ep <= max(1e-4, 2.0 * et):
            bad[k] = (ep, et)
    print("f64-arbitrated gradients: worst product %.2e (torch f32 %.2e) at %s" % (worst[1], worst[2], worst[0]))
    assert not bad, bad


# BASELINE.json configs[4]: the bts_test.py path (bts_test.py:84-128: model.eval(), torch.no_grad(), five outputs) at 704x1216,
# DenseNet161 widths.  The no-grad decoder runs the four LPG heads as fused chain kernels (reduction_1x1 + plane + LPG in one
# launch, lpg_chain_fwd_kernel<., 128|64|32, ., 8|4|2|1>); the checker is the oracle's formulas in f32 on the device.  Batch 4
# of the 32 (the per-image arithmetic does not depend on the batch size; 32 is what bench.py --mode infer times).
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_decoder_parity_inference_c5(dt):
    from bts_amd import profiler
    from bts_amd.model import bts
    B, H, W, feat, nf = 4, 704, 1216, DN161, 512
    gen = torch.Generator().manual_seed(505)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "kitti")
    Pd = {k: v.to(DEV) for k, v in P.items()}
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        ref, upd = O.decoder_forward(Pd, [f.to(DEV) for f in feats], focal.to(DEV), 80.0, "kitti", False)
    assert not upd                                               # eval mode: running statistics untouched
    dec = bts(NS(max_depth=80.0, dataset="kitti", encoder="densenet161_bts", bts_size=nf, decoder_dtype=dt), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).eval()
    before = {k: v.clone() for k, v in dec.state_dict().items()}
    prof = profiler.enable()
    with torch.no_grad():
        outs = dec([f.to(DEV) for f in feats], focal.to(DEV))
    names = [r[0] for r in prof.records]
    profiler.disable()
    # the fused inference heads ran for all four scales, and nothing was taped
    for k in (8, 4, 2, 1):
        assert "lpg_head_chain_fwd<k=%d>" % k in names, names
    assert not any(n.startswith(("lpg_head_fwd", "lpg_head_bwd", "lpg_head_chain_bwd")) for n in names), names
    assert all(o.shape == (B, 1, H, W) and o.dtype == torch.float32 for o in outs)
    rep = {"out%d" % i: (rel(o, r), l2rel(o, r)) for i, (o, r) in enumerate(zip(outs, ref))}
    print("parity c5 %s: (max, l2) %s" % (dt, {k: "%.1e %.1e" % v for k, v in rep.items()}))
    if dt == torch.float32:
        assert max(v[0] for v in rep.values()) < 1e-4, rep       # north_star bound, max-norm
    else:
        assert max(v[1] for v in rep.values()) < 1e-2, rep       # bf16 activation storage: bounded, not a parity claim
        # AbsRel of the final depth against the f32 checker (the figure bench.py --mode infer reports against the CPU oracle)
        absrel = ((outs[4] - ref[4]).abs() / ref[4]).mean().item()
        assert absrel < 1e-2, absrel
    for k, v in dec.state_dict().items():                        # eval: no buffer was touched
        assert torch.equal(v, before[k]), k
    # the forward has no atomics: bit-deterministic
    with torch.no_grad():
        again = dec([f.to(DEV) for f in feats], focal.to(DEV))
    assert all(torch.equal(a, b) for a, b in zip(outs, again))


def _elem_rel(a, b):
    """ELEMENT-WISE |a - b| / |b| (every output is a positive depth map): max over all pixels, and where it sits."""
    r = (a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)
    j = int(r.flatten().argmax().item())
    return r.flatten()[j].item(), j


def test_decoder_outputs_elementwise_1e4_full_c3_batch_f64_arbiter():
    """north_star's bound as written -- "outputs within 1e-4 relative" -- applied ELEMENT-WISE to every pixel of the five f32 outputs
    at the FULL configs[2] per-GPU batch (8 x 352 x 1216, DenseNet161 widths, train-mode BatchNorm over the 8 images), against the
    oracle's formulas in f64 on the device (forward only: the f64 convolutions of a backward pass over 8 images would take
    minutes).  torch's own f32 evaluation of the same formulas is measured beside it: where a pixel of the product is further
    than 1e-4 from f64, torch-f32 must be comparably far at that same pixel (bts.py:146 divides by n1 u + n2 v + n3; a pixel
    whose denominator is near zero is ill-conditioned for ANY f32 evaluation) -- otherwise the kernel is wrong."""
    from bts_amd.model import bts
    B, H, W, ds, md, feat, nf = 8, 352, 1216, "kitti", 80.0, DN161, 512
    gen = torch.Generator().manual_seed(4242)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, ds).to(DEV)
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        P64 = {k: (v.to(DEV).double() if v.dtype.is_floating_point else v.to(DEV)) for k, v in P.items()}
        o64, _ = O.decoder_forward(P64, [f.to(DEV).double() for f in feats], focal.double(), md, ds, True)
        o64 = [o.clone() for o in o64]
        del P64
        torch.cuda.empty_cache()
        P32 = {k: v.to(DEV) for k, v in P.items()}
        o32, _ = O.decoder_forward(P32, [f.to(DEV) for f in feats], focal, md, ds, True)
    dec = bts(NS(max_depth=md, dataset=ds, encoder="densenet161_bts", bts_size=nf, decoder_dtype=torch.float32), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).train()
    outs = dec([f.to(DEV).requires_grad_(True) for f in feats], focal)        # the recorded (training) forward: the timed kernels
    rep = []
    for i in range(5):
        ep, j = _elem_rel(outs[i], o64[i])
        et, _ = _elem_rel(o32[i], o64[i])
        et_same = ((o32[i].double().flatten()[j] - o64[i].flatten()[j]).abs() / o64[i].flatten()[j].abs().clamp_min(1e-30)).item()
        n_over = int((((outs[i].double() - o64[i]).abs() / o64[i].abs().clamp_min(1e-30)) > 1e-4).sum().item())
        rep.append((i, ep, et, et_same, n_over))
    print("element-wise vs f64, full C3 batch: " + "; ".join("out%d product %.2e torch-f32 %.2e (same pixel %.2e) over-1e-4: %d"
                                                             % r for r in rep))
    for i, ep, et, et_same, n_over in rep:
        assert ep < 1e-4 or ep <= 2.0 * et_same, (i, ep, et, et_same, n_over)
        assert n_over <= 8, (i, n_over)                    # of 3.4 M pixels per output


def test_decoder_densenet121_widths_c1_shape_vs_oracle():
    """BASELINE.json configs[0]'s decoder -- densenet121 widths [64, 64, 128, 256, 1024], 1 x 416 x 544, nyu -- in f32 on the HIP
    kernels against the CPU oracle: train-mode forward + backward and the eval-mode no-grad forward.  (The DenseNet161 / ResNeXt
    widths have their own full-size tests; these widths exercise other tile tails: conv3 sees 128 + 64 + 1 channels, conv2 64 + 64 + 1,
    upconv5 1024 -> 512.)"""
    from bts_amd.model import bts, silog_loss
    feat, nf, B, H, W = [64, 64, 128, 256, 1024], 512, 1, 416, 544
    gen = torch.Generator().manual_seed(121)
    P = O.make_decoder_params(feat, nf, gen, randomize_bn=True)
    feats = O.make_features(feat, B, H, W, gen)
    focal = O.synth_focal(B, "nyu")
    gt = O.synth_depth_gt(B, H, W, "nyu", gen)
    Pr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in P.items()}
    fr = [f.clone().requires_grad_(True) for f in feats]
    ref, upd = O.decoder_forward(Pr, fr, focal, 10.0, "nyu", True)
    lref = O.silog(ref[4], gt, gt > 0.1, 0.85)
    (lref + sum((o * o).mean() for o in ref[:4])).backward()
    dec = bts(NS(max_depth=10.0, dataset="nyu", encoder="densenet121_bts", bts_size=nf), feat, nf)
    dec.load_state_dict(P)
    dec.to(DEV).train()
    fs = [f.to(DEV).requires_grad_(True) for f in feats]
    outs = dec(fs, focal.to(DEV))
    loss = silog_loss(0.85)(outs[4], gt.to(DEV), (gt > 0.1).to(DEV))
    (loss + sum((o * o).mean() for o in outs[:4])).backward()
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert tuple(o.shape) == (B, 1, H, W)
        assert rel(o.cpu(), r) < 1e-4, i
    assert abs(loss.item() - lref.item()) / lref.item() < 1e-5
    for k_, v in upd.items():
        assert rel(dec.state_dict()[k_].cpu(), v) < 1e-4, k_
    bad = {}
    for n, p in dec.named_parameters():
        e = l2rel(p.grad.cpu(), Pr[n].grad)
        if not e < 5e-3:                     # same bound as the full-size f32 test upstream of the ASPP ReLUs; 1e-4 class downstream
            bad[n] = e
    for i, (f, g) in enumerate(zip(fs, fr)):
        e = l2rel(f.grad.cpu(), g.grad)
        if not e < 5e-3:
            bad["feat%d" % i] = e
    assert not bad, bad
    dec.eval()
    Pe = {k: v.detach().cpu() for k, v in dec.state_dict().items()}        # running statistics as the train step above left them
    with torch.no_grad():
        ref_e, _ = O.decoder_forward(Pe, feats, focal, 10.0, "nyu", False)
        out_e = dec([f.to(DEV) for f in feats], focal.to(DEV))
    for o, r in zip(out_e, ref_e):
        assert rel(o.cpu(), r) < 1e-4
