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
