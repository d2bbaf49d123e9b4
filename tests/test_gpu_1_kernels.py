"""GPU parity tests, kernel level: every HIP kernel family against the CPU oracle / plain PyTorch
f32 CPU math on the same seeded inputs, through the C ABI (bts_amd.ops / bts_amd.conv).

Tolerances: f32 paths 1e-4 relative (north_star); the LPG op itself is expected bit-exact for
integer-valued offsets (checked to 1e-6 and reported); bf16 paths 2e-2 relative (bf16 has 8
mantissa bits; stated here, not a parity claim).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bts_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def bf16_excess(a, b, half_ulps=2.0):
    """ELEMENT-WISE check of a result that was STORED as bf16 against the f32 reference computed from identical (bf16-rounded)
    operands: a correct kernel rounds the f32-accumulated value once, so |a - b| <= 2^-9 |b| per element; `half_ulps` = 2 allows
    2^-8 |b| (one more half-ulp for accumulation order).  Returns max(|a - b| - half_ulps * 2^-9 * |b|) / max|b| -- the part of the
    worst element's error that rounding cannot explain, relative to the tensor's scale; a dropped tap or border row is O(1e-1)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    ex = ((a - b).abs() - half_ulps * 2.0 ** -9 * b.abs()).clamp_min(0.0).max()
    return (ex / b.abs().max().clamp_min(1e-30)).item()


def bf16_bad(a, b, half_ulps=2.0, atol_rel=1e-4):
    """number of elements outside |a - b| <= half_ulps * 2^-9 |b| + atol_rel * max|b| (for results behind a ReLU mask: an input
    within f32 rounding of zero may take the other branch in two correct evaluations -- a handful of elements per million)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return int(((a - b).abs() > half_ulps * 2.0 ** -9 * b.abs() + atol_rel * b.abs().max()).sum().item())


BF16_ATOL = 1e-4      # of max|b|: f32 accumulation-order noise of a K <= 10^4 contraction is ~1e-6 of it


def load(golden_dir, name):
    return np.load("%s/%s.npz" % (golden_dir, name))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [8, 4, 2])
def test_lpg_op_golden(golden_dir, k):
    """local_planar_guidance fwd/bwd vs the reference's own outputs (tests/golden/lpg.npz)."""
    from bts_amd.model import local_planar_guidance
    g = load(golden_dir, "lpg")
    eq = torch.tensor(g["k%d_eq" % k], device=DEV, requires_grad=True)
    out = local_planar_guidance(k)(eq, torch.ones(eq.shape[0], device=DEV))
    ref = torch.tensor(g["k%d_out" % k])
    assert out.shape == ref.shape
    exact = (out.cpu() == ref).float().mean().item()
    print("lpg k=%d bit-exact fraction %.4f rel %.3e" % (k, exact, rel(out, ref)))
    assert rel(out, ref) < 1e-6
    out.backward(torch.tensor(g["k%d_gout" % k], device=DEV))
    assert rel(eq.grad, torch.tensor(g["k%d_geq" % k])) < 1e-4


@pytest.mark.parametrize("shape", [(2, 16, 24), (8, 64, 224)], ids=["small", "row_split"])
def test_lpg_op_multi_matches_single_launches(shape):
    """bts_lpg_fwd_multi / bts_lpg_bwd_multi: the k = 8 / 4 / 2 problems of one batch in ONE launch -- bit-identical to the three
    single launches (same kernel bodies, blocks dealt by range), at a size where each cell is one thread and at a size where the
    k = 8 / 4 problems split a cell's patch rows over several threads (the backward then joins the row partials through LDS); and
    the golden vectors of the reference through the multi entry."""
    from bts_amd import ops
    B, H, W = shape
    gen = torch.Generator().manual_seed(B * 1000 + W)
    ks = [8, 4, 2]
    eqs = [torch.randn(B, H // k, W // k, 4, generator=gen).to(DEV) for k in ks]
    for e in eqs:
        e[..., 2] += 3.0                                  # keep the denominators away from zero
    divs = [80.0, 1.0, 10.0]
    outs = ops.lpg_fwd_multi(eqs, ks, divs)
    for e, k, d, o in zip(eqs, ks, divs, outs):
        assert torch.equal(o, ops.lpg_fwd(e, k, d))
    gs = [torch.randn(B, H, W, generator=gen).to(DEV) for _ in ks]
    geqs = ops.lpg_bwd_multi(gs, eqs, ks, divs)
    for g, e, k, d, ge in zip(gs, eqs, ks, divs, geqs):
        assert torch.equal(ge, ops.lpg_bwd(g, e, k, d))
    # two problems only, different batch sizes
    o2 = ops.lpg_fwd_multi([eqs[2][:1].contiguous(), eqs[0]], [2, 8])
    assert torch.equal(o2[0], ops.lpg_fwd(eqs[2][:1].contiguous(), 2)) and torch.equal(o2[1], ops.lpg_fwd(eqs[0], 8))


def test_lpg_op_multi_golden(golden_dir):
    from bts_amd import ops
    g = load(golden_dir, "lpg")
    ks = [8, 4, 2]
    eqs = [torch.tensor(g["k%d_eq" % k], device=DEV).permute(0, 2, 3, 1).contiguous() for k in ks]     # golden: NCHW; the op: [B][h][w][4]
    outs = ops.lpg_fwd_multi(eqs, ks)
    geqs = ops.lpg_bwd_multi([torch.tensor(g["k%d_gout" % k], device=DEV) for k in ks], eqs, ks)
    for k, o, ge in zip(ks, outs, geqs):
        assert rel(o, torch.tensor(g["k%d_out" % k])) < 1e-6
        assert rel(ge.permute(0, 3, 1, 2), torch.tensor(g["k%d_geq" % k])) < 1e-4


@pytest.mark.parametrize("k", [8, 4, 2])
def test_lpg_head_vs_oracle(k):
    from bts_amd import ops
    gen = torch.Generator().manual_seed(100 + k)
    B, h, w = 2, 7, 9
    raw = torch.randn(B, h, w, 4, generator=gen)
    raw_c = raw.permute(0, 3, 1, 2)[:, :3].clone().requires_grad_(True)
    eq = O.normalize_plane(O.plane_from_raw(raw_c, 80.0))
    ref = O.lpg(eq, k) / 80.0
    depth, plane = ops.lpg_head_fwd(raw.to(DEV), k, 80.0, want_plane=True)
    assert rel(depth, ref) < 1e-4
    assert rel(plane, eq.permute(0, 2, 3, 1)) < 1e-5
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
        pad = 4 if dt == torch.float32 else 8
        graw = ops.lpg_head_bwd(raw.to(DEV), gy.to(DEV), k, 80.0, dt, pad)
        assert rel(graw[..., :3].float(), raw_c.grad.permute(0, 2, 3, 1)) < tol
        assert graw[..., 3:].abs().max().item() == 0.0


@pytest.mark.parametrize("tag", ["kitti", "nyu"])
def test_silog_golden(golden_dir, tag):
    from bts_amd.model import silog_loss
    g = load(golden_dir, "silog")
    est = torch.tensor(g[tag + "_est"], device=DEV, requires_grad=True)
    gt = torch.tensor(g[tag + "_gt"], device=DEV)
    mask = gt > (1.0 if tag == "kitti" else 0.1)
    loss = silog_loss(float(g[tag + "_vf"]))(est, gt, mask)
    assert loss.dim() == 0
    assert abs(loss.item() - float(g[tag + "_loss"])) / float(g[tag + "_loss"]) < 1e-5
    loss.backward()
    assert rel(est.grad, torch.tensor(g[tag + "_gest"])) < 1e-4


def test_silog_ragged_and_empty_mask():
    """n not a multiple of 4, and a mask selecting a single pixel (edge cases of the reduction)."""
    from bts_amd.model import silog_loss
    gen = torch.Generator().manual_seed(5)
    est = torch.rand(1, 1, 7, 9, generator=gen) * 5 + 0.5
    gt = torch.rand(1, 1, 7, 9, generator=gen) * 5 + 0.5
    mask = torch.rand(1, 1, 7, 9, generator=gen) > 0.5
    ref = O.silog(est, gt, mask, 0.85)
    got = silog_loss(0.85)(est.to(DEV), gt.to(DEV), mask.to(DEV))
    assert abs(got.item() - ref.item()) / ref.item() < 1e-5


def test_silog_unaligned_views():
    """est / gt / mask as already-contiguous views with an odd storage offset (a sliced batch with odd H*W): the vectorised
    kernels need 16-byte aligned pointers, the wrappers copy instead of failing."""
    from bts_amd.model import silog_loss
    gen = torch.Generator().manual_seed(6)
    est = torch.rand(3, 1, 7, 9, generator=gen) * 5 + 0.5
    gt = torch.rand(3, 1, 7, 9, generator=gen) * 5 + 0.5
    mask = torch.rand(3, 1, 7, 9, generator=gen) > 0.4
    e, g, m = est.to(DEV)[1:].requires_grad_(True), gt.to(DEV)[1:], mask.to(DEV)[1:]
    assert e.is_contiguous() and e.data_ptr() % 16 != 0
    ec = est[1:].clone().requires_grad_(True)
    ref = O.silog(ec, gt[1:], mask[1:], 0.85)
    ref.backward()
    got = silog_loss(0.85)(e, g, m)
    got.backward()
    assert abs(got.item() - ref.item()) / ref.item() < 1e-5
    assert rel(e.grad.cpu(), ec.grad) < 1e-4


# ------------------------------------------------------------------------------------------------
def _nhwc(x, dt, v):
    """NCHW f32 cpu -> NHWC device tensor with channels zero-padded to a multiple of v."""
    N, C, H, W = x.shape
    cp = (C + v - 1) // v * v
    t = torch.zeros(N, H, W, cp, dtype=dt, device=DEV)
    t[..., :C] = x.permute(0, 2, 3, 1).to(DEV).to(dt)
    return t


CONV_CASES = [
    # name, cout, seg_channels, kk, dil, up, (N,H,W)
    ("c3x3_2seg", 40, [16, 24], 9, 1, False, (2, 9, 13)),
    ("c3x3_slot", 16, [8, 8, 1], 9, 1, False, (2, 8, 12)),
    ("c3x3_big", 136, [64, 40], 9, 1, False, (1, 12, 20)),
    ("dil6", 24, [32], 9, 6, False, (2, 10, 14)),
    ("dil24", 16, [16], 9, 24, False, (1, 13, 17)),
    ("c1x1_wide", 64, [72], 1, 1, False, (2, 6, 10)),
    ("c1x1_to3", 3, [8], 1, 1, False, (2, 5, 7)),
    ("c3x3_to1", 1, [16], 9, 1, False, (2, 8, 8)),
    ("c3x3_to1_ragged", 1, [32, 8], 9, 1, False, (3, 11, 37)),
    # >= 256 tiles of 8x32 pixels: the bf16 weight gradient takes the LDS-halo kernel (conv_wgrad_halo)
    ("wg_halo_conv1like", 32, [32, 8], 9, 1, False, (8, 61, 125)),
    ("wg_halo_conv2like", 64, [64, 96, 8], 9, 1, False, (8, 61, 125)),
    ("wg_halo_1group", 24, [32], 9, 1, False, (8, 64, 128)),
    ("wide96_dgrad", 128, [96, 64], 9, 1, False, (8, 61, 125)),   # data gradient towards a 96-channel skip: conv_halo_wide<3> (96-co tiles)
    ("wide96_fwd", 192, [64, 72], 9, 1, False, (8, 61, 125)),    # 192 = 2 x 96 output channels: the forward on 96-co tiles
    ("wg_halo_wide_co", 136, [64, 40], 9, 1, False, (8, 61, 125)),  # > 64 output channels on a >= 256-tile map (3 ragged co tiles)
    ("wg_halo_up1like", 32, [64], 9, 1, True, (8, 61, 125)),      # conv_wgrad_halo_up, ragged
    ("wg_halo_up2like", 64, [128], 9, 1, True, (8, 64, 128)),     # 2 ci groups x 2 co groups
    ("wg_halo_up_small", 16, [24], 9, 1, True, (8, 64, 128)),     # one ci tile per role
    ("up_small", 16, [24], 9, 1, True, (2, 5, 6)),
    ("up_big", 72, [136], 9, 1, True, (1, 6, 9)),
    ("daspp6seg", 32, [16, 8, 8, 8, 8, 8], 9, 1, False, (1, 9, 11)),
    # channel counts that are multiples of 64: conv_igemm_dma runs channel-chunk-major (taps inner)
    ("kmajor_2seg", 136, [64, 64], 9, 1, False, (1, 12, 20)),
    ("kmajor_dil", 72, [128], 9, 6, False, (2, 10, 14)),
    ("kmajor_up", 72, [128], 9, 1, True, (1, 6, 9)),
    ("kmajor_3seg_wide", 130, [64, 128, 64], 9, 1, False, (2, 7, 9)),
    # wide bf16 weight gradients (conv_wgrad_tr: LDS-DMA + transposing reads): many pixel chunks, pixel splits joined by
    # atomics, ragged Cout / K, sub-pixel phases, dilation, a 1x1 over a 3-segment concat
    ("wtr_multi", 200, [64, 72], 9, 1, False, (2, 31, 45)),
    ("wtr_up", 136, [128], 9, 1, True, (2, 17, 23)),
    ("wtr_1x1", 256, [64, 128, 64], 1, 1, False, (2, 20, 31)),
    ("wtr_dil", 128, [256], 9, 12, False, (1, 29, 40)),
    # weights in MFMA A-fragment order + resident pixel tile (conv_igemm_res): > 64 output rows, <= 4 chunks of K, on launches that
    # always take the implicit GEMM: ragged Cout over two output tiles with a half-filled last chunk, 3 + 1 row tiles, several
    # input segments, the four-phase forward of an up-convolution (4 taps x 64 channels = 4 chunks per phase), many output tiles
    ("res_up_4ch", 72, [64], 9, 1, True, (2, 9, 13)),
    ("res_1x1_ragged", 200, [96], 1, 1, False, (2, 17, 19)),
    ("res_1x1_3seg_5tiles", 600, [64, 128, 64], 1, 1, False, (2, 13, 21)),
    ("res_1x1_dgrad_wide", 256, [576], 1, 1, False, (2, 13, 21)),
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dt):
    """Implicit-GEMM conv (forward, data-gradient, weight-gradient) vs F.conv2d autograd on CPU f32."""
    from bts_amd.conv import ConvLayer
    from bts_amd import ops
    from bts_amd._lib import ACT_NONE
    name, cout, segc, kk, dil, up, (N, H, W) = case
    v = 4 if dt == torch.float32 else 8
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    cin = sum(segc)
    xs = [torch.randn(N, c, H, W, generator=gen) for c in segc]
    k = 3 if kk == 9 else 1
    w = torch.randn(cout, cin, k, k, generator=gen) * (1.0 / (cin * kk) ** 0.5)
    if dt == torch.bfloat16:   # compare against the same bf16-rounded operands
        xs = [x.to(dt).float() for x in xs]
    xs_r = [x.clone().requires_grad_(True) for x in xs]
    w_r = w.clone().requires_grad_(True)
    xin = torch.cat(xs_r, 1)
    if up and dt != torch.bfloat16:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    # (the bf16-rounded weight is the autograd LEAF: differentiating through `.to(bf16).float()` rounds the weight gradient to bf16 on
    # its way back, and the comparison would measure that rounding -- 3.4e-3 on every case, gpurun r05a -- instead of the kernel)
    if dt == torch.bfloat16:
        w_r = w.to(dt).float().requires_grad_(True)
    wq = w_r
    if up and dt == torch.bfloat16:
        # nearest x2 + 3x3 (bts.py:76-79) as the product evaluates it: four 2x2 sub-pixel phases on the LOW-resolution input with the
        # 3x3 taps that land on the same low-res pixel summed in f32 and THEN rounded to bf16 (one rounding per phase weight instead of
        # one per tap: the product's operand is bf16(sum), and a tap-wise rounded reference differs from it by 2^-9 per weight --
        # 2e-3 on the output, gpurun r05b).  Written from the definition: output row 2i + a reads up-sampled rows 2i + a + ky - 1,
        # i.e. low-res row i + floor((a + ky - 1) / 2).  Straight-through rounding keeps the weight gradient that of the f32 sum.
        wf = w.clone().requires_grad_(True)                        # un-rounded master weight: the leaf of the weight gradient
        w_r = wf
        Hl, Wl = xin.shape[2], xin.shape[3]
        xp = F.pad(xin, (1, 1, 1, 1))
        ref = torch.zeros(N, cout, 2 * Hl, 2 * Wl)
        rows = []
        for a_ in (0, 1):
            grp = {}
            for ky in range(3):
                grp.setdefault((a_ + ky - 1) // 2, []).append(ky)
            rows.append(sorted(grp.items()))                       # [(low-res offset, [ky ...]), (offset + 1, [...])]
        outs_ph = {}
        for a_ in (0, 1):
            for b_ in (0, 1):
                wp_ = torch.stack([torch.stack([sum(wf[:, :, ky, kx] for ky in kys for kx in kxs) for _, kxs in rows[b_]], -1)
                                   for _, kys in rows[a_]], -2)            # [co][ci][2][2]
                wp_ = wp_ + (wp_.detach().to(dt).float() - wp_.detach())
                o = F.conv2d(xp, wp_)                                       # [N][co][Hl+1][Wl+1]: window starts at padded (i, j)
                oy, ox = rows[a_][0][0] + 1, rows[b_][0][0] + 1            # first offset -1 -> padded start i, 0 -> i + 1
                outs_ph[(a_, b_)] = o[:, :, oy:oy + Hl, ox:ox + Wl]
        ref = torch.stack([torch.stack([outs_ph[(a_, 0)], outs_ph[(a_, 1)]], -1) for a_ in (0, 1)], -3)   # [N][co][Hl][2][Wl][2]
        ref = ref.reshape(N, cout, 2 * Hl, 2 * Wl)
    else:
        ref = F.conv2d(xin, wq, padding=dil if kk == 9 else 0, dilation=dil)
    gy = torch.randn(ref.shape, generator=gen)
    if dt == torch.bfloat16:
        gy = gy.to(dt).float()
    ref.backward(gy)

    L = ConvLayer(name, cout, segc, kk, dil, up)
    wd_dev = w.to(DEV)
    segs = [_nhwc(x, dt, v) for x in xs]
    Ho, Wo = ref.shape[2], ref.shape[3]
    cp = (cout + v - 1) // v * v
    out = torch.zeros(N, Ho, Wo, cp, dtype=dt, device=DEV)
    L.forward(segs, L.pack_fwd(wd_dev, dt), out, ACT_NONE)
    torch.cuda.synchronize()
    bf = dt == torch.bfloat16

    def err(a, b, half_ulps=2.0):
        """f32: max-norm relative error (bar 1e-4); bf16-stored results: the element-wise excess over bf16 rounding (bar BF16_ATOL)"""
        return bf16_excess(a, b, half_ulps) if bf else rel(a, b)
    tol = BF16_ATOL if bf else 1e-4
    e = err(out[..., :cout].float().permute(0, 3, 1, 2), ref)
    print("%s %s fwd err %.3e (max-norm %.3e)" % (name, dt, e, rel(out[..., :cout].float().permute(0, 3, 1, 2), ref)))
    assert e < tol, "fwd"
    assert out[..., cout:].abs().max().item() == 0.0 if cp > cout else True
    # the same launch with the weights in fragment order (global -> VGPR, conv_igemm_res): same chunks, same k-steps, same MFMA
    # sequence per accumulator as conv_igemm_dma -- bit-identical, with and without an activation epilogue
    tb = L.tables(dt, torch.device(DEV))
    lay = L.frag_layout(dt, cout, tb["ktot"])
    if lay:
        from bts_amd._lib import ACT_ELU
        wpf = L.to_frag(L.pack_fwd(wd_dev, dt), L._launch_taps(False)[0])
        for act in (ACT_NONE, ACT_ELU):
            o1, o2 = torch.zeros_like(out), torch.zeros_like(out)
            L.forward(segs, L.pack_fwd(wd_dev, dt), o1, act)
            L.forward(segs, wpf, o2, act, w_frag=lay)
            assert torch.equal(o1, o2), "fragment-order forward (layout %d, act %d)" % (lay, act)
        _plog("res_fwd", c0=0, k=0, dt=str(dt), name=name, layout=lay, l2=0.0, max=0.0)

    dz = _nhwc(gy, dt, v)
    for i, (x, c) in enumerate(zip(xs_r, segc)):
        gx = torch.empty_like(segs[i])
        L.dgrad(dz, L.pack_dgrad(wd_dev, dt, i), i, gx, False)
        layd = L.frag_layout(dt, segs[i].shape[3], tb["cout_pad"], True)
        if layd:
            wdf = L.to_frag(L.pack_dgrad(wd_dev, dt, i), L._launch_taps(True)[0])
            g2 = torch.empty_like(gx)
            L.dgrad(dz, wdf, i, g2, False, w_frag=layd)
            assert torch.equal(g2, gx), "fragment-order dgrad seg %d (layout %d)" % (i, layd)
            yv0 = _nhwc(torch.randn(x.shape, generator=torch.Generator().manual_seed(5)), dt, v)
            b0 = _nhwc(torch.randn(x.shape, generator=torch.Generator().manual_seed(6)), dt, v)
            ga1, ga2 = b0.clone(), b0.clone()
            L.dgrad(dz, L.pack_dgrad(wd_dev, dt, i), i, ga1, True, yv0)
            L.dgrad(dz, wdf, i, ga2, True, yv0, w_frag=layd)
            assert torch.equal(ga1, ga2), "fragment-order dgrad fold+acc seg %d" % i
            _plog("res_dgrad", c0=0, k=0, dt=str(dt), name=name, layout=layd, seg=i, l2=0.0, max=0.0)
        e = err(gx[..., :c].float().permute(0, 3, 1, 2), x.grad)
        print("%s %s dgrad seg%d err %.3e" % (name, dt, i, e))
        assert e < tol, "dgrad seg %d" % i
        # accumulate mode adds on top (bf16: the first result was rounded, the sum is rounded again: 3 half-ulps of 2 g)
        L.dgrad(dz, L.pack_dgrad(wd_dev, dt, i), i, gx, True)
        assert err(gx[..., :c].float().permute(0, 3, 1, 2), 2 * x.grad, 3.0) < 2 * tol
        # ELU-derivative fold (bts_conv_desc_t::fold_elu_y): the launch that completes the gradient of an ELU output takes it
        # through the ELU -- (result [+ old]) * (y > 0 ? 1 : y + 1) -- in every kernel's epilogue, written and accumulated
        yv = _nhwc(torch.randn(x.shape, generator=gen), dt, v)
        fac = torch.where(yv.float() > 0, torch.ones_like(yv, dtype=torch.float32), yv.float() + 1.0)[..., :c].permute(0, 3, 1, 2).cpu()
        gf = torch.empty_like(segs[i])
        L.dgrad(dz, L.pack_dgrad(wd_dev, dt, i), i, gf, False, yv)
        assert err(gf[..., :c].float().permute(0, 3, 1, 2), x.grad * fac) < tol, "fold seg %d" % i
        base = _nhwc(torch.randn(x.shape, generator=gen), dt, v)
        ga = base.clone()
        L.dgrad(dz, L.pack_dgrad(wd_dev, dt, i), i, ga, True, yv)
        want = (x.grad + base[..., :c].float().permute(0, 3, 1, 2).cpu()) * fac
        assert err(ga[..., :c].float().permute(0, 3, 1, 2), want) < 2 * tol, "fold+acc seg %d" % i
    # the weight gradient is f32 on both paths, from identical (bf16-rounded) operands: f32 accumulation order is all that differs
    gw = L.wgrad(segs, dz)
    e = rel(gw, w_r.grad)
    print("%s %s wgrad rel %.3e" % (name, dt, e))
    assert e < 1e-4, "wgrad"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 9, 13), (1, 32, 21, 131), (3, 64, 7, 70)], ids=["c16", "c32_ragged", "c64"])
def test_conv3x3_c1_streaming_kernels(dt, shape):
    """get_depth's dedicated kernels (csrc/conv_c1.hip): sigmoid(conv3x3 to one channel) * scale[n], and the data gradient with the
    sigmoid derivative formed from (grad_y, y), written / accumulated / folded through an ELU output -- against torch autograd."""
    from bts_amd import ops
    from bts_amd._lib import BtsAmdError
    N, Cc, H, W = shape
    gen = torch.Generator().manual_seed(N * 100 + Cc)
    v = 4 if dt == torch.float32 else 8
    x = torch.randn(N, Cc, H, W, generator=gen)
    w = torch.randn(1, Cc, 3, 3, generator=gen) * (1.0 / (9 * Cc) ** 0.5)
    if dt == torch.bfloat16:
        x = x.to(dt).float()
    wq = w.to(dt).float()
    sc_n = torch.tensor([1.01, 0.97, 1.05][:N])
    xr = x.clone().requires_grad_(True)
    yr = torch.sigmoid(F.conv2d(xr, wq, padding=1)) * 80.0 * sc_n.view(-1, 1, 1, 1)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xd = _nhwc(x, dt, v)
    assert ops.conv_c1_supported(xd)
    wd, scd = w.to(DEV), sc_n.to(DEV)
    assert ops.conv_c1_supported(xd, wd) and not ops.conv_c1_supported(xd, wd[:, :Cc - 1].contiguous())
    with pytest.raises(BtsAmdError):         # the kernels index w by x's (padded) channel count: a narrower weight must not reach them
        ops.conv3x3_c1_fwd(xd, wd[:, :Cc - 1].contiguous(), 80.0, scd)
    y = ops.conv3x3_c1_fwd(xd, wd, 80.0, scd)
    assert rel(y.unsqueeze(1), yr) < (1e-5 if dt == torch.float32 else 1e-4)     # f32 output on both paths, identical bf16 operands
    gyd = gy.squeeze(1).to(DEV).contiguous()
    gx = torch.full(xd.shape, float("nan"), dtype=dt, device=DEV)
    ops.conv3x3_c1_dgrad(gyd, y, wd, gx, False, 80.0, scd)
    bf = dt == torch.bfloat16
    tol = BF16_ATOL if bf else 1e-4

    def err(a, b, half_ulps=2.0):
        return bf16_excess(a, b, half_ulps) if bf else rel(a, b)
    assert err(gx.float().permute(0, 3, 1, 2), xr.grad) < tol
    base = _nhwc(torch.randn(x.shape, generator=gen), dt, v)
    yelu = _nhwc(torch.randn(x.shape, generator=gen), dt, v)
    fac = torch.where(yelu.float() > 0, torch.ones_like(yelu, dtype=torch.float32), yelu.float() + 1.0).permute(0, 3, 1, 2).cpu()
    ga = base.clone()
    ops.conv3x3_c1_dgrad(gyd, y, wd, ga, True, 80.0, scd, yelu)
    want = (xr.grad + base.float().permute(0, 3, 1, 2).cpu()) * fac
    assert err(ga.float().permute(0, 3, 1, 2), want) < 2 * tol
    gf = torch.empty_like(xd)
    ops.conv3x3_c1_dgrad(gyd, y, wd, gf, False, 80.0, None, yelu)       # no per-image scale: y was produced with one, so only shape-check
    assert torch.isfinite(gf.float()).all()
    # weight gradient into the packed [1][9][Ktot] layout (accumulating: starts from a known base), Ktot wider than C
    wq_r = wq.clone().requires_grad_(True)
    (torch.sigmoid(F.conv2d(x, wq_r, padding=1)) * 80.0 * sc_n.view(-1, 1, 1, 1)).backward(gy)
    ktot = Cc + v
    dwp = torch.full((1, 9, ktot), 0.5, device=DEV)
    ops.conv3x3_c1_wgrad(gyd, y, xd, dwp, 80.0, scd)
    got = (dwp[0, :, :Cc] - 0.5).t().reshape(Cc, 3, 3).cpu()            # [t][c] -> [c][ky][kx]
    assert rel(got, wq_r.grad[0]) < 2e-4                                  # f32 sums on both paths
    assert (dwp[0, :, Cc:] == 0.5).all()                                # the padding columns are not touched


@pytest.mark.parametrize("shape", [(2, 21, 70), (8, 61, 125)], ids=["small", "multi_tile"])
@pytest.mark.parametrize("mode", ["plain", "acc_fold"])
def test_conv_dual_data_gradient(shape, mode):
    """conv1-like layer (32 + 4 -> 32 channels, bf16): the data gradients towards both input tensors from ONE launch
    (bts_conv_desc_t::y2) against F.conv2d autograd and against the two single launches (bit-identical: same MFMA order)."""
    from bts_amd.conv import ConvLayer
    N, H, W = shape
    dt, v = torch.bfloat16, 8
    gen = torch.Generator().manual_seed(N * 1000 + W)
    cout, segc = 32, [32, 4]
    w = torch.randn(cout, sum(segc), 3, 3, generator=gen) * (1.0 / (36 * 9) ** 0.5)
    wq = w.to(dt).float()
    xs = [torch.randn(N, c, H, W, generator=gen).to(dt).float().requires_grad_(True) for c in segc]
    dz = torch.randn(N, cout, H, W, generator=gen).to(dt).float()
    F.conv2d(torch.cat(xs, 1), wq, padding=1).backward(dz)
    L = ConvLayer("c1like", cout, segc, 9)
    assert L.dual_dgrad_ok(dt, 0, 1)
    wd = [L.pack_dgrad(w.to(DEV), dt, i) for i in range(2)]
    dzd = _nhwc(dz, dt, v)
    fold = _nhwc(torch.randn(N, 32, H, W, generator=gen), dt, v) if mode == "acc_fold" else None
    base0 = _nhwc(torch.randn(N, 32, H, W, generator=gen), dt, v)
    base1 = _nhwc(torch.randn(N, 4, H, W, generator=gen), dt, v)
    acc = mode == "acc_fold"
    g0 = base0.clone() if acc else torch.full_like(base0, float("nan"))
    g1 = base1.clone() if acc else torch.full_like(base1, float("nan"))
    L.dgrad_dual(dzd, wd[0], 0, g0, acc, fold, wd[1], 1, g1, acc)
    r0 = base0.clone() if acc else torch.full_like(base0, float("nan"))
    r1 = base1.clone() if acc else torch.full_like(base1, float("nan"))
    L.dgrad(dzd, wd[0], 0, r0, acc, fold)
    L.dgrad(dzd, wd[1], 1, r1, acc)
    assert torch.equal(g0, r0) and torch.equal(g1, r1)
    want0, want1 = xs[0].grad, xs[1].grad
    if acc:
        fac = torch.where(fold.float() > 0, torch.ones_like(fold, dtype=torch.float32), fold.float() + 1.0).permute(0, 3, 1, 2).cpu()
        want0 = (want0 + base0.float().permute(0, 3, 1, 2).cpu()) * fac
        want1 = want1 + base1.float().permute(0, 3, 1, 2).cpu()[:, :4]
    assert bf16_excess(g0.float().permute(0, 3, 1, 2), want0) < BF16_ATOL
    assert bf16_excess(g1.float().permute(0, 3, 1, 2)[:, :4], want1) < BF16_ATOL
    assert g1[..., 4:].float().abs().max().item() == (base1[..., 4:].float().abs().max().item() if acc else 0.0)


def test_conv_wgrad_group():
    """bts_conv_wgrad_group: five independent weight gradients (dense-ASPP-like 1x1 and dilated 3x3 layers, one multi-segment 3x3,
    different widths) in one launch of the ring kernel, against the per-layer launches (same kernel body: equal up to the order of
    the split-K atomics) and against F.conv2d autograd."""
    from bts_amd.conv import ConvLayer
    dt, v = torch.bfloat16, 8
    N, H, W = 4, 44, 76
    gen = torch.Generator().manual_seed(77)
    specs = [("g1x1a", 256, [320], 1, 1), ("g3x3d6", 128, [256], 9, 6), ("g1x1b", 256, [256, 192, 128], 1, 1),
             ("g3x3d3", 128, [256], 9, 3), ("g3x3", 160, [128, 64], 9, 1)]
    items, refs, singles = [], [], []
    for name, cout, segc, kk, dil in specs:
        L = ConvLayer(name, cout, segc, kk, dil)
        assert L.wgrad_groupable(dt, N, H, W), name
        xs = [torch.randn(N, c, H, W, generator=gen).to(dt).float() for c in segc]
        k = 3 if kk == 9 else 1
        w = (torch.randn(cout, sum(segc), k, k, generator=gen) * 0.05).requires_grad_(True)
        dz = torch.randn(N, cout, H, W, generator=gen).to(dt).float()
        F.conv2d(torch.cat(xs, 1), w, padding=dil if kk == 9 else 0, dilation=dil).backward(dz)
        segs = [_nhwc(x, dt, v) for x in xs]
        dzd = _nhwc(dz, dt, v)
        tb = L.tables(dt, torch.device(DEV))
        dwp = torch.zeros((cout, L.T, tb["ktot"]), dtype=torch.float32, device=DEV)
        items.append((L, segs, dzd, dwp))
        refs.append(w.grad)
        one = torch.zeros_like(dwp)
        L.wgrad_packed(segs, dzd, one)
        singles.append(one)
    ConvLayer.wgrad_group(items)
    for (L, segs, dzd, dwp), ref, one in zip(items, refs, singles):
        assert rel(dwp, one) < 1e-5, L.name
        assert rel(L.unpack_wgrad(dwp, dt).cpu(), ref) < 1e-4, L.name        # f32 sums from identical bf16-rounded operands
    # fewer than five, and a second call accumulates on top
    ConvLayer.wgrad_group(items[:2])
    for (L, segs, dzd, dwp), one in zip(items[:2], singles[:2]):
        assert rel(dwp, 2 * one) < 1e-5, L.name


def test_conv_epilogues():
    """ELU / sigmoid*scale_n epilogues and the single-channel f32 map output."""
    from bts_amd.conv import ConvLayer
    from bts_amd._lib import ACT_ELU, ACT_SIGMOID
    gen = torch.Generator().manual_seed(3)
    N, H, W = 2, 8, 12
    x = torch.randn(N, 16, H, W, generator=gen)
    w = torch.randn(24, 16, 3, 3, generator=gen) * 0.1
    L = ConvLayer("e", 24, [16], 9)
    seg = [_nhwc(x, torch.float32, 4)]
    out = torch.empty(N, H, W, 24, device=DEV)
    L.forward(seg, L.pack_fwd(w.to(DEV), torch.float32), out, ACT_ELU)
    assert rel(out.permute(0, 3, 1, 2), F.elu(F.conv2d(x, w, padding=1))) < 1e-4
    w1 = torch.randn(1, 16, 3, 3, generator=gen) * 0.1
    L1 = ConvLayer("g", 1, [16], 9)
    sc = torch.tensor([1.01, 0.99], device=DEV)
    m = torch.empty(N, H, W, device=DEV)
    L1.forward(seg, L1.pack_fwd(w1.to(DEV), torch.float32), m, ACT_SIGMOID, 80.0, sc)
    ref = 80.0 * torch.sigmoid(F.conv2d(x, w1, padding=1)) * sc.cpu().view(-1, 1, 1, 1)
    assert rel(m.unsqueeze(1), ref) < 1e-4


STATS_CASES = [
    # name, cout, seg_channels, kk, dil, up, (N,H,W), act is ELU, frag layout, rows expected (None: some, 0: no statistics epilogue)
    ("up_elu_128", 128, [96], 9, 1, True, (2, 9, 13), True, False, None),          # conv_igemm_dma 128 x 128, ELU stores, four phases
    ("up_elu_256_ragged", 256, [64, 72], 9, 1, True, (1, 11, 23), True, False, None),
    ("up_elu_536rows", 128, [32], 9, 1, True, (2, 61, 70), True, False, None),      # 536 partial rows, ragged
    ("c1x1_res", 256, [96, 64], 1, 1, False, (2, 17, 19), False, True, None),      # conv_igemm_res, ragged last pixel tile
    ("c1x1_dma_longk", 128, [576], 1, 1, False, (2, 13, 21), False, False, None),  # 9 chunks: not the resident form
    ("dil_64", 64, [128], 9, 12, False, (2, 21, 29), False, False, None),          # 64 x 128 tiles, four wave columns
    ("dil_32", 32, [64], 9, 6, False, (1, 23, 31), False, False, None),            # 32 x 256 tiles
    ("halo_no_stats", 64, [64], 9, 1, False, (2, 9, 13), True, False, 0),          # conv_halo: no statistics epilogue -> bn_stats
    ("padded_no_stats", 72, [64], 1, 1, False, (2, 9, 13), False, False, 0),       # Cout % 32 != 0
]


@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_conv_epilogue_batch_statistics(case):
    """bts_conv_desc_t::stats_ws: the batch statistics a convolution's epilogue forms of its own (stored, bf16) output are those of a
    bn_stats pass over that output -- same values summed, only the f32 partial-sum grouping differs -- and the output itself is
    bit-identical to the launch without them."""
    from bts_amd.conv import ConvLayer
    from bts_amd import ops
    from bts_amd._lib import ACT_ELU, ACT_NONE
    name, cout, segc, kk, dil, up, (N, H, W), elu, frag, rows_expected = case
    dt = torch.bfloat16
    gen = torch.Generator().manual_seed(sum(map(ord, name)))
    cin = sum(segc)
    xs = [torch.randn(N, c, H, W, generator=gen) + 0.3 for c in segc]       # non-zero channel means
    k = 3 if kk == 9 else 1
    w = torch.randn(cout, cin, k, k, generator=gen) * (1.0 / (cin * kk) ** 0.5)
    L = ConvLayer(name, cout, segc, kk, dil, up)
    segs = [_nhwc(x, dt, 8) for x in xs]
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    cp = (cout + 7) // 8 * 8
    act = ACT_ELU if elu else ACT_NONE
    wp, lay = L.pack_fwd(w.to(DEV), dt), 0
    if frag:
        lay = L.frag_layout(dt, cout, L.tables(dt, torch.device(DEV))["ktot"])
        assert lay
        wp = L.to_frag(wp, L._launch_taps(False)[0])
    o0 = torch.zeros(N, Ho, Wo, cp, dtype=dt, device=DEV)
    o1 = torch.zeros_like(o0)
    L.forward(segs, wp, o0, act, w_frag=lay)
    st = []
    L.forward(segs, wp, o1, act, w_frag=lay, stats=st)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1)
    if rows_expected == 0:
        assert st == [] and list(L._stats_rows.values()) == [0]
        return
    assert len(st) == 1 and all(r > 0 for r in L._stats_rows.values())
    mean, var = st[0]
    m_ref, v_ref = ops.bn_stats(o1)
    xf = o1.double()
    m64 = xf.mean((0, 1, 2))
    v64 = xf.var((0, 1, 2), unbiased=False)
    e2 = (xf * xf).mean((0, 1, 2))
    # f32 partial sums over <= 256 values, combined in double: the mean to 1e-6 of the rms, the variance (E[x^2] - mean^2) likewise
    em = ((mean.double() - m64).abs() / e2.sqrt()).max().item()
    ev = ((var.double() - v64).abs() / e2).max().item()
    em_ref = ((m_ref.double() - m64).abs() / e2.sqrt()).max().item()
    ev_ref = ((v_ref.double() - v64).abs() / e2).max().item()
    print("%s epilogue stats vs f64: mean %.2e var %.2e (bn_stats pass: %.2e %.2e), rows %s" % (name, em, ev, em_ref, ev_ref,
                                                                                             list(L._stats_rows.values())))
    _plog("conv_epilogue_stats", c0=cout, k=kk, dt=str(dt), name=name, l2=ev, max=em)
    assert em < 2e-6 and ev < 2e-6


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(3, 40, 9, 11), (2, 72, 97, 103)], ids=["small", "multi_iter"])
def test_batchnorm_train_fwd_bwd(dt, relu, shape):
    """bn_stats + bn_prepare + affine_act and the two-pass backward vs F.batch_norm autograd (CPU f32).  `multi_iter`:
    ~20 k pixels on 9 / 18 channel vectors, so every thread of the streaming kernels walks several pixels (the batched
    4-iteration main loop and its tail), with a pixel count that is not a multiple of anything."""
    from bts_amd import ops
    from bts_amd._lib import ACT_NONE, ACT_RELU
    gen = torch.Generator().manual_seed(7)
    N, C, H, W = shape
    v = 4 if dt == torch.float32 else 8
    bf = dt == torch.bfloat16
    x = torch.randn(N, C, H, W, generator=gen) * 1.5 + 0.3
    if bf:
        x = x.to(dt).float()
    g = torch.rand(C, generator=gen) + 0.5
    b = torch.rand(C, generator=gen) - 0.5
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.batch_norm(xr, rm, rv, gr, br, True, 0.01, 1.1e-5)
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=gen)
    if bf:
        gy = gy.to(dt).float()          # identical operands on both sides: what differs is the ONE rounding of each stored result
    y.backward(gy)

    def close(a_, b_, what):
        """f32: 1e-4 max-norm (5e-4 for the gradients: two reductions deep); bf16-stored: element-wise 2^-8 |b| + 1e-4 max|b|, with
        <= 3 elements per tensor allowed on the other side of a ReLU mask"""
        if not bf:
            assert rel(a_, b_) < (1e-4 if what == "y" else 5e-4), what
        else:
            assert bf16_bad(a_, b_) <= (3 if relu else 0), (what, bf16_bad(a_, b_), bf16_excess(a_, b_))

    xt = _nhwc(x, dt, v)
    mean, var = ops.bn_stats(xt)
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    invstd, scale, shift = ops.bn_prepare(mean, var, N * H * W, g.to(DEV), b.to(DEV), 1.1e-5, 0.01, rm_d, rv_d)
    yt = ops.affine_act(xt, scale, shift, ACT_RELU if relu else ACT_NONE)
    close(yt.float().permute(0, 3, 1, 2), y, "y")
    assert rel(rm_d, rm) < 1e-4 and rel(rv_d, rv) < 1e-4
    dx = torch.empty_like(xt)
    db, dg = ops.bn_bwd(_nhwc(gy, dt, v), xt, mean, invstd, g.to(DEV), b.to(DEV), relu, dx, False)
    close(dx.float().permute(0, 3, 1, 2), xr.grad, "dx")
    assert rel(dg, gr.grad) < 5e-4 and rel(db, br.grad) < 5e-4          # f32 sums on both paths
    # accumulate form (a tensor that feeds several BatchNorms of the dense ASPP): dx += ..., bit-identical sums
    base = torch.randn(dx.shape, generator=gen).to(dt).to(DEV)
    acc = base.clone()
    db2, dg2 = ops.bn_bwd(_nhwc(gy, dt, v), xt, mean, invstd, g.to(DEV), b.to(DEV), relu, acc, True)
    assert torch.equal(db2, db) and torch.equal(dg2, dg)
    assert rel(acc.float() - base.float(), dx.float()) < (1e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["cat3_relu", "cat6_relu_acc", "single_elu_fold", "single_relu_copy", "odd_vectors"])
@pytest.mark.parametrize("train", [True, False], ids=["train", "eval"])
def test_batchnorm_concat_multi_segment(dt, case, train):
    """bts_bn_apply / bts_bn_bwd: BatchNorm(+ReLU) over a channel concatenation in one launch (the dense-ASPP first_bn layers,
    bts.py:51-66) against F.batch_norm autograd on the materialised torch.cat (CPU f32): output, running statistics, every
    segment's input gradient (written or accumulated), dgamma / dbeta; with the producer's ELU derivative folded into the
    gradient (x = ELU output of a convolution, bts.py:199-208) and with the extra relu(y) output of bn4_2 (bts.py:208-210)."""
    from bts_amd import ops
    gen = torch.Generator().manual_seed(23)
    v = 4 if dt == torch.float32 else 8
    tol = 1e-4 if dt == torch.float32 else 2e-2
    N, H, W = 2, 37, 41
    chans, relu, fold, copy, accs = {
        "cat3_relu": ([64, 48, 32], True, False, False, [False, True, False]),
        "cat6_relu_acc": ([64, 48, 32, 32, 32, 32], True, False, False, [True, True, False, False, True, False]),
        "single_elu_fold": ([72], False, True, False, [False]),
        "single_relu_copy": ([40], False, True, True, [False]),
        "odd_vectors": ([24, 40], True, False, False, [False, False]),     # 3 / 5 (6 / 10) channel vectors: guarded tails
    }[case]
    ctot = sum(chans)
    eps = 1.1e-5
    zs = [(torch.randn(N, c, H, W, generator=gen) * 1.5 + 0.3) for c in chans]
    if dt == torch.bfloat16:
        zs = [z.to(dt).float() for z in zs]
    g = torch.rand(ctot, generator=gen) + 0.5
    b = torch.rand(ctot, generator=gen) - 0.5
    rm, rv = torch.rand(ctot, generator=gen) - 0.5, torch.rand(ctot, generator=gen) + 0.5
    # ---- reference: x = ELU(z) for the folded case (z is the convolution's accumulator), cat, batch_norm, relu
    zr = [z.clone().requires_grad_(True) for z in zs]
    xr = [F.elu(z) for z in zr] if fold else zr
    if fold and dt == torch.bfloat16:      # the stored activation is the bf16-rounded ELU output
        xr = [x + (x.detach().to(dt).float() - x.detach()) for x in xr]
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(torch.cat(xr, 1), rm_ref, rv_ref, gr, br, train, 0.01, eps)
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=gen)
    bf = dt == torch.bfloat16
    if bf:
        gy = gy.to(dt).float()          # identical operands on both sides
    obj = (y * gy).sum()
    if copy:
        gy2 = torch.randn(y.shape, generator=gen)
        if bf:
            gy2 = gy2.to(dt).float()
        obj = obj + (F.relu(y) * gy2).sum()
    obj.backward()
    # ---- product
    xt = [_nhwc(x.detach(), dt, v) for x in xr]
    gd, bd, rmd, rvd = g.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV)
    if train:
        stats = [ops.bn_stats(x) for x in xt]
    else:
        stats, c0 = [], 0
        for c in chans:
            stats.append((rmd[c0:c0 + c], rvd[c0:c0 + c]))
            c0 += c
    out = torch.empty(N, H, W, ctot, dtype=dt, device=DEV)
    out2 = torch.empty_like(out) if copy else None
    ops.bn_apply(xt, stats, gd, bd, eps, relu, out, out2, 0.01 if train else 0.0, rmd if train else None, rvd if train else None)
    if bf:      # stored once: element-wise 2^-8 |y| + 1e-4 max|y|
        assert bf16_bad(out.float().permute(0, 3, 1, 2), y) <= (3 if relu else 0), bf16_excess(out.float().permute(0, 3, 1, 2), y)
    else:
        assert rel(out.float().permute(0, 3, 1, 2), y) < tol
    if copy:
        assert torch.equal(out2, torch.relu(out))
    assert rel(rmd, rm_ref) < 1e-4 and rel(rvd, rv_ref) < 1e-4
    dy = _nhwc(gy, dt, v)
    if copy:      # the ReLU copy's backward accumulates into the gradient of y (decoder: act_bwd(..., accumulate=True))
        from bts_amd._lib import ACT_RELU
        ops.act_bwd(_nhwc(gy2, dt, v), out2, ACT_RELU, out=dy, accumulate=True)
    bases = [torch.randn(x.shape, generator=gen).to(dt).to(DEV) if a else None for x, a in zip(xt, accs)]
    dxs = [bs.clone() if a else torch.empty_like(x) for x, a, bs in zip(xt, accs, bases)]
    db, dg = ops.bn_bwd_ms(dy, xt, dxs, accs, stats, gd, bd, eps, relu, train, fold)
    c0 = 0
    for i, (z, dx, a, bs) in enumerate(zip(zr, dxs, accs, bases)):
        if not bf:
            got = dx.float() - bs.float() if a else dx.float()
            assert rel(got.permute(0, 3, 1, 2), z.grad) < tol * (10 if a else 5), (case, i)
        elif fold or copy:
            # the ELU factor comes from the STORED (bf16) ELU output, x + 1 against autograd's exp(z): 2^-9 |x| apart; the ReLU-copy
            # gradient is summed into dy in bf16 first (a second rounding of the incoming gradient): bounded, not rounding-exact
            assert rel(dx.float().permute(0, 3, 1, 2), z.grad) < 1e-2, (case, i)
        else:
            want = z.grad + (bs.float().permute(0, 3, 1, 2).cpu() if a else 0.0)      # the stored value: (old +) gradient, rounded once
            assert bf16_bad(dx.float().permute(0, 3, 1, 2), want) <= (3 if relu else 0), (case, i, bf16_excess(dx.float().permute(0, 3, 1, 2), want))
        c0 += z.shape[1]
    assert rel(dg, gr.grad) < (5e-4 if not (bf and (fold or copy)) else 1e-2) and rel(db, br.grad) < (5e-4 if not (bf and copy) else 1e-2)


def _clear_relu_borderline(gy, x, g, b, rm, rv, train, eps, c0):
    """Zero gy[:, c0:c0+C] where the BatchNorm output (f64 evaluation) is within 1e-5 of the ReLU threshold: two correct f32
    evaluations of y may take different branches there -- ONE such element (y = 1.7e-9 in f64, -1.4e-8 in torch's CPU kernel,
    gradient 1.6) moved dbeta of a 2 494-pixel channel by 8 % in the first run of the test below."""
    xd = x.double()
    m = xd.mean((0, 2, 3)) if train else rm.double()
    v = xd.var((0, 2, 3), unbiased=False) if train else rv.double()
    y = (xd - m[None, :, None, None]) / torch.sqrt(v + eps)[None, :, None, None] * g.double()[None, :, None, None] + b.double()[None, :, None, None]
    C = x.shape[1]
    gy[:, c0:c0 + C][y.abs() < 1e-5] = 0.0
    return gy


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("relu,acc,train", [(True, False, True), (True, True, True), (False, False, True), (True, True, False)],
                         ids=["relu", "relu_acc", "plain", "relu_acc_eval"])
def test_batchnorm_shared_input_multi_backward(dt, n, relu, acc, train):
    """bts_bn_bwd_multi: n BatchNorm(+ReLU) layers that normalise the SAME tensor with the same batch statistics and their own
    gamma / beta (the dense-ASPP first_bn layers all see cat(up4, skip2, daspp_3, ...), bts.py:51-66 + 211-218), backward in one
    reduction + one apply pass, against F.batch_norm autograd of the n layers on CPU f32: the summed input gradient (written or
    accumulated), every dgamma / dbeta.  The output gradients are channel SLICES of wider tensors, as in the decoder."""
    from bts_amd import ops
    gen = torch.Generator().manual_seed(31 + n)
    v = 4 if dt == torch.float32 else 8
    bf = dt == torch.bfloat16
    N, H, W, Cc, eps = 2, 29, 43, 40, 1.1e-5
    x = torch.randn(N, Cc, H, W, generator=gen) * 1.5 + 0.3
    if bf:
        x = x.to(dt).float()
    xr = x.clone().requires_grad_(True)
    gs = [(torch.rand(Cc, generator=gen) + 0.5) for _ in range(n)]
    bs_ = [(torch.rand(Cc, generator=gen) - 0.5) for _ in range(n)]
    rm, rv = torch.rand(Cc, generator=gen) - 0.5, torch.rand(Cc, generator=gen) + 0.5
    grs = [g.clone().requires_grad_(True) for g in gs]
    brs = [b.clone().requires_grad_(True) for b in bs_]
    wide = [Cc + 8 * (i + 1) for i in range(n)]          # BatchNorm i's concatenation is wider; x sits at channel offset 8
    gys, obj = [], 0.0
    for i in range(n):
        y = F.batch_norm(xr, rm.clone(), rv.clone(), grs[i], brs[i], train, 0.01, eps)
        if relu:
            y = F.relu(y)
        gy = torch.randn(N, wide[i], H, W, generator=gen)
        if bf:
            gy = gy.to(dt).float()
        if relu:
            gy = _clear_relu_borderline(gy, x, gs[i], bs_[i], rm, rv, train, eps, 8)
        gys.append(gy)
        obj = obj + (y * gy[:, 8:8 + Cc]).sum()
    obj.backward()
    xt = _nhwc(x, dt, v)
    if train:
        mean, var = ops.bn_stats(xt)
    else:
        mean, var = rm.to(DEV), rv.to(DEV)
    base = torch.randn(xt.shape, generator=gen).to(dt).to(DEV)
    dx = base.clone() if acc else torch.empty_like(xt)
    contribs, keep, dyws = [], [], []
    for i in range(n):
        dyw = _nhwc(gys[i], dt, v)
        gw, bw = torch.zeros(wide[i], device=DEV), torch.zeros(wide[i], device=DEV)
        gw[8:8 + Cc], bw[8:8 + Cc] = gs[i].to(DEV), bs_[i].to(DEV)
        sums = torch.full((2, wide[i]), float("nan"), device=DEV)
        keep.append(sums)
        dyws.append(dyw)
        contribs.append((dyw[..., 8:8 + Cc], gw[8:8 + Cc], bw[8:8 + Cc], sums[0, 8:8 + Cc], sums[1, 8:8 + Cc]))
    # a second tensor in the same launch (the n = 4 group of the decoder has three: up3, skip, daspp_3): 24 channels -- another lane
    # width than the first one alone would take --, accumulating iff the first does not, its own statistics and parameter slices
    gen2 = torch.Generator().manual_seed(131 + n)
    C2 = 24
    x2 = torch.randn(N, C2, H, W, generator=gen2) * 0.7 - 0.2
    if bf:
        x2 = x2.to(dt).float()
    x2r = x2.clone().requires_grad_(True)
    g2 = [(torch.rand(C2, generator=gen2) + 0.5) for _ in range(n)]
    b2 = [(torch.rand(C2, generator=gen2) - 0.5) for _ in range(n)]
    rm2, rv2 = torch.rand(C2, generator=gen2) - 0.5, torch.rand(C2, generator=gen2) + 0.5
    g2r = [g.clone().requires_grad_(True) for g in g2]
    b2r = [b.clone().requires_grad_(True) for b in b2]
    gy2, obj2 = [], 0.0
    for i in range(n):
        y2 = F.batch_norm(x2r, rm2.clone(), rv2.clone(), g2r[i], b2r[i], train, 0.01, eps)
        if relu:
            y2 = F.relu(y2)
        t = torch.randn(N, C2, H, W, generator=gen2)
        if bf:
            t = t.to(dt).float()
        if relu:
            t = _clear_relu_borderline(t, x2, g2[i], b2[i], rm2, rv2, train, eps, 0)
        gy2.append(t)
        obj2 = obj2 + (y2 * t).sum()
    obj2.backward()
    x2t = _nhwc(x2, dt, v)
    mean2, var2 = ops.bn_stats(x2t) if train else (rm2.to(DEV), rv2.to(DEV))
    base2 = torch.randn(x2t.shape, generator=gen2).to(dt).to(DEV)
    dx2t = base2.clone() if not acc else torch.empty_like(x2t)
    sums2 = [torch.full((2, C2), float("nan"), device=DEV) for _ in range(n)]
    contribs2 = [(_nhwc(gy2[i], dt, v), g2[i].to(DEV), b2[i].to(DEV), sums2[i][0], sums2[i][1]) for i in range(n)]
    ops.bn_bwd_multi([(xt, mean, var, dx, acc, contribs), (x2t, mean2, var2, dx2t, not acc, contribs2)], eps, relu, train)
    torch.cuda.synchronize()
    for i in range(n):
        assert rel(sums2[i][1], g2r[i].grad) < 5e-4 and rel(sums2[i][0], b2r[i].grad) < 5e-4, ("second tensor", i)
    want2 = x2r.grad + (base2.float().permute(0, 3, 1, 2).cpu() if not acc else 0.0)
    got2 = dx2t.float().permute(0, 3, 1, 2)
    if bf:
        assert bf16_bad(got2, want2) <= 3 * n, bf16_excess(got2, want2)
    else:
        assert rel(got2, want2) < 1e-4 * (10 if not acc else 5)
    def device_sums(i):      # the kernel's own arithmetic with torch ops on the device (diagnostic for a failing comparison)
        xh = (xt.float() - mean) * torch.rsqrt(var + eps)
        d = dyws[i][..., 8:8 + Cc].float()
        if relu:
            d = torch.where(xh * gs[i].to(DEV) + bs_[i].to(DEV) > 0, d, torch.zeros_like(d))
        return d.sum((0, 1, 2)).cpu(), (d * xh).sum((0, 1, 2)).cpu()
    for i in range(n):      # the statistics first: a wrong sum shows up here before it does in dx
        eg, eb = rel(keep[i][1, 8:8 + Cc], grs[i].grad), rel(keep[i][0, 8:8 + Cc], brs[i].grad)
        if not (eg < 5e-4 and eb < 5e-4):
            db_dev, dg_dev = device_sums(i)
            worst = int((keep[i][0, 8:8 + Cc].cpu() - brs[i].grad).abs().argmax())
            raise AssertionError("BatchNorm %d: dgamma %.3g dbeta %.3g; worst channel %d: kernel %.6f, device torch %.6f, cpu autograd %.6f; "
                                 "beta %.6f gamma %.6f" % (i, eg, eb, worst, keep[i][0, 8 + worst].item(), db_dev[worst].item(),
                                                          brs[i].grad[worst].item(), bs_[i][worst].item(), gs[i][worst].item()))
        assert torch.isnan(keep[i][:, :8]).all() and torch.isnan(keep[i][:, 8 + Cc:]).all()      # nothing outside the channel range
    want = xr.grad + (base.float().permute(0, 3, 1, 2).cpu() if acc else 0.0)
    got = dx.float().permute(0, 3, 1, 2)
    if bf:
        assert bf16_bad(got, want) <= 3 * n, bf16_excess(got, want)
    else:
        assert rel(got, want) < 1e-4 * (10 if acc else 5)
    # the same through one bts_bn_bwd per BatchNorm (the rounds 3-5 form): the two product paths agree to f32 summation order
    if train and not bf:
        dx2 = base.clone() if acc else torch.zeros_like(xt)
        for i in range(n):
            ops.bn_bwd_ms(dyws[i][..., 8:8 + Cc], [xt], [dx2], [True], [(mean, var)], gs[i].to(DEV), bs_[i].to(DEV), eps, relu, True)
        assert rel(dx.float(), dx2.float()) < 2e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_act_bwd_accumulate(dt):
    from bts_amd import ops
    from bts_amd._lib import ACT_ELU, ACT_RELU, BtsAmdError
    gen = torch.Generator().manual_seed(12)
    N, H, W, C = 2, 53, 61, 40
    y = torch.randn(N, H, W, C, generator=gen).to(dt).to(DEV)
    gy = torch.randn(N, H, W, C, generator=gen).to(dt).to(DEV)
    base = torch.randn(N, H, W, C, generator=gen).to(dt).to(DEV)
    for act in (ACT_ELU, ACT_RELU):
        yf, gf = y.float(), gy.float()
        d = (gf * torch.where(yf > 0, torch.ones_like(yf), yf + 1.0)) if act == ACT_ELU else torch.where(yf > 0, gf, torch.zeros_like(gf))
        acc = base.clone()
        ops.act_bwd(gy, y, act, out=acc, accumulate=True)
        assert torch.equal(acc, (d + base.float()).to(dt))
    with pytest.raises(BtsAmdError):        # in-place accumulate is meaningless and refused
        ops.act_bwd(gy, y, ACT_ELU, out=gy, accumulate=True)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_act_bwd_streaming(dt):
    """ELU' / ReLU' through the vector kernel, out of place and in place (the decoder's form), several pixels per thread."""
    from bts_amd import ops
    from bts_amd._lib import ACT_ELU, ACT_RELU
    gen = torch.Generator().manual_seed(11)
    N, H, W, C = 2, 97, 103, 72
    y = (torch.randn(N, H, W, C, generator=gen)).to(dt).to(DEV)
    gy = torch.randn(N, H, W, C, generator=gen).to(dt).to(DEV)
    for act in (ACT_ELU, ACT_RELU):
        yf, gf = y.float(), gy.float()
        ref = (gf * torch.where(yf > 0, torch.ones_like(yf), yf + 1.0)) if act == ACT_ELU else torch.where(yf > 0, gf, torch.zeros_like(gf))
        ref = ref.to(dt)
        out = ops.act_bwd(gy, y, act)
        assert torch.equal(out, ref)
        g2 = gy.clone()
        ops.act_bwd(g2, y, act, out=g2)
        assert torch.equal(g2, ref)


def test_layout_wide_bf16_ragged_tiles():
    """bf16 -> bf16 conversions on the 64x64-tile kernels (H*W % 8 == 0, C % 8 == 0): 7 full pixel tiles + 8 pixels, 3 full
    channel tiles + 8 channels, padded NHWC pitch; exact both ways, pad channels untouched."""
    from bts_amd import ops
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(2, 200, 19, 24, generator=gen).to(torch.bfloat16).to(DEV)
    t = ops.nchw_to_nhwc(x, torch.bfloat16, c_pad=208)
    assert torch.equal(t[..., :200].permute(0, 3, 1, 2), x) and t[..., 200:].abs().max().item() == 0
    back = ops.nhwc_to_nchw(t, 200, out_dtype=torch.bfloat16)
    assert torch.equal(back, x)


def test_layout_roundtrip_and_pack_maps():
    from bts_amd import ops
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 24, 7, 10, generator=gen)
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
        t = ops.nchw_to_nhwc(x.to(DEV), dt, relu=True)
        assert rel(t.float().permute(0, 3, 1, 2), F.relu(x)) <= tol
        back = ops.nhwc_to_nchw(t, 24, relu_src=x.to(DEV))
        tb = ops.nchw_to_nhwc(x.to(DEV).to(torch.bfloat16), dt, relu=True)        # bf16 source (autocast encoder)
        assert rel(tb.float().permute(0, 3, 1, 2), F.relu(x)) <= 8e-3
        assert rel(back, F.relu(x)) <= tol
    d8 = torch.randn(2, 16, 24, generator=gen).to(DEV)
    slot = ops.pack_maps([d8], [4], 2, 4, 6, torch.float32)
    assert torch.equal(slot[..., 0], d8[:, ::4, ::4]) and slot[..., 1:].abs().max().item() == 0
    maps = [torch.randn(2, 8, 8, generator=gen).to(DEV) for _ in range(4)]
    slot = ops.pack_maps(maps, [1, 1, 1, 1], 2, 8, 8, torch.bfloat16)
    for s in range(4):
        assert rel(slot[..., s].float(), maps[s]) < 8e-3
    g = torch.randn(2, 8, 8, 8, generator=gen).to(DEV)
    gm = [torch.zeros(2, 8, 8, device=DEV) for _ in range(4)]
    ops.unpack_maps(g, gm, [1, 1, 1, 1])
    for s in range(4):
        assert torch.equal(gm[s], g[..., s])


def _plog(tag, **vals):
    """measured parity figures -> gpurun_out/parity_bounds.jsonl (so that one GPU run shows how far inside its bound every check is)"""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_bounds.jsonl"), "a") as f:
            f.write(json.dumps(dict(tag=tag, **{k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in vals.items()})) + "\n")
    except OSError:
        pass


def _chain_reference(x, ws, k, md, gy, base, fold, rounding):
    """The reduction chain + head + LPG and its backward on the CPU.  rounding = True restates the bf16 kernel's ROUNDING POINTS
    (csrc/lpg_chain.hip): every ELU output, every dz and the stored dx are rounded to bf16 once, the ELU derivative is taken from the
    rounded ELU output (y > 0 ? 1 : y + 1), contractions accumulate in f32 (here f64, rounded to f32), the head runs in f32 on the
    un-rounded accumulator of the last layer -- so kernel and reference differ only by f32 accumulation order and by the rare bf16
    rounding decisions that order flips.  rounding = False is plain f32 autograd of the same chain (what the f32 kernel matches).
    x: [cells, c0] (bf16-representable), ws: [co, ci] (bf16-representable when rounding), gy like the head output.
    Returns (head output, dx [cells, c0], [dW_l])."""
    bf = (lambda t: t.bfloat16().float()) if rounding else (lambda t: t)
    mm = lambda a_, b_: (a_.double() @ b_.double()).float()
    elu = (lambda z: torch.maximum(z, torch.exp(z.clamp(max=0.0)) - 1.0)) if rounding else F.elu
    acts = [x.float()]
    zs = []
    for w in ws[:-1]:
        zs.append(mm(acts[-1], w.t()))
        acts.append(bf(elu(zs[-1])))
    raw = mm(acts[-1], ws[-1].t()).requires_grad_(True)                     # [cells, 3 or 1]
    B, h, w_ = gy.shape[0], (gy.shape[1] // k if k > 1 else gy.shape[1]), (gy.shape[2] // k if k > 1 else gy.shape[2])
    r4 = raw.t().reshape(1, raw.shape[1], B * h, w_).reshape(raw.shape[1], B, h, w_).permute(1, 0, 2, 3)
    out = (O.lpg(O.normalize_plane(O.plane_from_raw(r4, md)), k) / md) if k > 1 else torch.sigmoid(r4[:, 0])
    out.backward(gy)
    dz = bf(raw.grad)
    dws = [None] * len(ws)
    dA = None
    for l in range(len(ws) - 1, -1, -1):
        dws[l] = mm(dz.t(), acts[l])
        dA = mm(dz, ws[l])
        if l > 0:
            y = acts[l]
            dfac = torch.where(y > 0, torch.ones_like(y), y + 1.0) if rounding else torch.where(zs[l - 1] > 0, torch.ones_like(y), torch.exp(zs[l - 1]))
            dz = bf(dA * dfac)
    dx = dA + (base if base is not None else 0.0)
    if fold:
        xf = x.float()
        dx = dx * torch.where(xf > 0, torch.ones_like(xf), xf + 1.0)
    return out.detach(), bf(dx), dws


@pytest.mark.parametrize("c0,k", [(64, 2), (32, 1), (16, 2), (64, 4), (16, 1), (32, 4), (64, 8), (16, 8), (128, 4), (128, 8)])
@pytest.mark.parametrize("acc", [False, True])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_lpg_chain_bwd_vs_autograd(c0, k, acc, dt):
    """Fused recompute backward of a reduction chain + head (csrc/lpg_chain.hip; bts.py:83-146 under autograd).
    f32: against plain f32 autograd of the chain, 1e-4 (north_star's bound).
    bf16: against a reference that restates the kernel's rounding points (_chain_reference): what is left is accumulation order, so
    the bounds are those of a few flipped bf16 roundings (1e-3 class), not of five layers of bf16 noise (the 0.10 / 0.25 this test
    carried before could not see a wrong tap or row).  Cell count is not a multiple of the 32-cell tile."""
    from bts_amd import chain
    bf = dt == torch.bfloat16
    if not chain.bwd_supported(c0, False, k, dt):
        pytest.skip("no fused backward for this (c0, k, dtype)")
    gen = torch.Generator().manual_seed(7 * c0 + k)
    B, h, w, md = 2, 13, 21, 80.0
    dims = [c0]
    while dims[-1] > 8:
        dims.append(dims[-1] // 2)
    dims.append(3 if k > 1 else 1)
    gain = 1.0 if c0 >= 128 else 2.0       # (He-scaled 128-wide chains drive the synthetic planes near-singular)
    rnd = (lambda t: t.bfloat16().float()) if bf else (lambda t: t)
    ws = [rnd(torch.randn(dims[i + 1], dims[i], 1, 1, generator=gen) * (gain / dims[i]) ** 0.5) for i in range(len(dims) - 1)]
    x = rnd(torch.randn(B, h, w, c0, generator=gen))
    gy = torch.randn((B, h * k, w * k) if k > 1 else (B, h, w), generator=gen)
    gx0 = rnd(torch.randn(B, h, w, c0, generator=gen))
    w2 = [wi.reshape(wi.shape[0], wi.shape[1]) for wi in ws]
    ref_out, ref_dx, ref_dw = _chain_reference(x.reshape(-1, c0), w2, k, md, gy, gx0.reshape(-1, c0) if acc else None, False, bf)
    _, ref_dx_fold, _ = _chain_reference(x.reshape(-1, c0), w2, k, md, gy, gx0.reshape(-1, c0) if acc else None, True, bf)

    def rel_l2(a_, b_):
        a_, b_ = a_.detach().double().cpu(), b_.detach().double().cpu()
        return ((a_ - b_).norm() / b_.norm()).item()
    wd = [wi.to(DEV) for wi in ws]
    frags, frags_t = chain.pack_chain(wd, dt), chain.pack_chain_t(wd, dt)
    xd = x.to(dt).to(DEV)
    out = chain.chain_fwd(xd, frags, c0, False, k, md)
    e_fwd, e_fwd2 = rel(out, ref_out.reshape(out.shape)), rel_l2(out, ref_out.reshape(out.shape))
    _plog("chain_fwd", c0=c0, k=k, dt=str(dt), max=e_fwd, l2=e_fwd2)
    # f32: 1e-4; bf16 vs the rounding-exact reference: the plane head amplifies the few flipped roundings (division by n1 u + n2 v + n3)
    assert (e_fwd2 < 1e-5 and e_fwd < 1e-4) if not bf else (e_fwd2 < 5e-3 and e_fwd < 5e-2), (e_fwd, e_fwd2)
    gx = gx0.to(dt).to(DEV) if acc else torch.full((B, h, w, c0), float("nan"), dtype=dt, device=DEV)
    gws = [torch.zeros(dims[i + 1], max(dims[i], 8), device=DEV) for i in range(len(dims) - 1)]
    gyd = gy.to(DEV).contiguous()
    chain.chain_bwd(xd, frags, frags_t, c0, k, md, gyd, gx, acc, gws)
    want = ref_dx.reshape(B, h, w, c0)
    e_l2, e_mx = rel_l2(gx.float(), want), rel(gx.float(), want)
    bad = bf16_bad(gx.float(), want, 2.0, 2e-3) if bf else 0
    _plog("chain_bwd_dx", c0=c0, k=k, acc=acc, dt=str(dt), l2=e_l2, max=e_mx, bad=bad, n=gx.numel())
    if bf:      # element-wise: 2^-8 |b| + 2e-3 max|b| for all but 0.2 % of the elements (a flipped rounding upstream moves a whole cell)
        assert e_l2 < 5e-3 and bad <= 2e-3 * gx.numel(), (e_l2, e_mx, bad)
    else:
        assert e_mx < 1e-4 and e_l2 < 1e-5, (e_l2, e_mx)
    for li, (g, rg) in enumerate(zip(gws, ref_dw)):
        ci = rg.shape[1]
        l2, mxv = rel_l2(g[:, :ci], rg), rel(g[:, :ci], rg)
        _plog("chain_bwd_dw", c0=c0, k=k, acc=acc, dt=str(dt), layer=li, l2=l2, max=mxv)
        assert (l2 < 2e-3 and mxv < 5e-3) if bf else (l2 < 1e-5 and mxv < 1e-4), (li, l2, mxv)
        assert g[:, ci:].abs().max().item() == 0.0 if g.shape[1] > ci else True
    # x_is_elu_output: the chain completes the gradient of its input, an ELU output, and takes it through the ELU:
    # (dx [+ old]) * (x > 0 ? 1 : x + 1), rounded once
    gx2 = gx0.to(dt).to(DEV) if acc else torch.full((B, h, w, c0), float("nan"), dtype=dt, device=DEV)
    gws2 = [torch.zeros_like(g) for g in gws]
    chain.chain_bwd(xd, frags, frags_t, c0, k, md, gyd, gx2, acc, gws2, True)
    wantf = ref_dx_fold.reshape(B, h, w, c0)
    if bf:
        assert rel_l2(gx2.float(), wantf) < 5e-3 and bf16_bad(gx2.float(), wantf, 2.0, 2e-3) <= 2e-3 * gx.numel()
    else:
        assert rel(gx2.float(), wantf) < 1e-4
    for g, g2 in zip(gws, gws2):
        assert rel(g2, g) < 1e-5          # same sums, atomics in a different order


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batched_pack_unpack_match_single_layer_kernels(dt):
    """bts_pack_weight_batch / bts_unpack_wgrad_batch (one launch for every decoder layer, LDS-tiled) must reproduce
    the per-layer bts_pack_weight / bts_unpack_wgrad bit for bit: forward operands, data-gradient operands of every
    input segment, and the weight gradients scattered back to PyTorch layout (odd channel counts, phases, dilations)."""
    from bts_amd.decoder import DecoderPlan, PackSet
    # (nf = 256: layers with more than 64 output rows exist, so the bf16 set also holds MFMA-fragment-order operands -- 1x1, dilated,
    # up-convolution forward / data-gradient in both K orders --, checked against ConvLayer.to_frag's plain-torch statement of it)
    feat, nf = [8, 24, 16, 40, 56], 256
    plan = DecoderPlan(feat, nf)
    gen = torch.Generator().manual_seed(11)
    P = {k: v.to(DEV) for k, v in O.make_decoder_params(feat, nf, gen).items()}
    ps = PackSet(plan, P, dt)
    ps.pack_forward()
    ps.pack_dgrad()
    dwp = torch.randn(ps.dwp_total, generator=gen).to(DEV)
    gw = torch.full((ps.gw_total,), float("nan"), device=DEV)
    ps.unpack_all(dwp, gw)
    nfrag = set()
    for n, L in plan.layers.items():
        w = P[n + ".weight"]
        want = L.pack_fwd(w, dt)
        if ps.fwd_frag[n]:
            want = L.to_frag(want, L._launch_taps(False)[0])
            nfrag.add(("fwd", L.kk))
        assert torch.equal(ps.fwd[n], want), n
        for i in range(len(L.seg_channels)):
            want = L.pack_dgrad(w, dt, i)
            if ps.dgrad_frag[(n, i)]:
                want = L.to_frag(want, L._launch_taps(True)[0])
                nfrag.add(("dgrad", L.kk))
            assert torch.equal(ps.dgrad[(n, i)], want), (n, i)
    if dt == torch.bfloat16:
        from bts_amd.conv import _res_enabled
        assert not _res_enabled() or {("fwd", 1), ("dgrad", 1)} <= nfrag, nfrag       # 1x1 layers with K <= 256: forward and data gradient
    else:
        assert not nfrag
    for n, L in plan.layers.items():
        off, shape = ps.dwp_off[n]
        goff, gshape = ps.gw_off[n]
        want = L.unpack_wgrad(dwp[off:off + shape[0] * shape[1] * shape[2]].view(shape), dt)
        assert torch.equal(gw[goff:goff + want.numel()].view(gshape), want), n


@pytest.mark.parametrize("tag", ["kitti", "nyu"])
def test_eval_errors_vs_reference_golden(golden_dir, tag):
    """bts_eval_errors (one device reduction per batch) against the reference's compute_errors on the same seeded maps
    (tests/golden/eval.npz), through online_eval's paste-back / clamp / crop semantics; a batch of 3 with one image
    flagged has_valid_depth = 0 and one all-invalid image checks the eval_measures accumulation (bts_main.py:258-299).
    Tolerance 2e-5 relative: sums are f64 here, pairwise f32 in numpy; log / log10 differ by <= 2 ulp."""
    import numpy as np
    from bts_amd import evalops
    from oracle import eval_oracle as E
    g = np.load("%s/eval.npz" % golden_dir)
    gh, gw, ph, pw, md, kb, garg, eig, ds = E.EVAL_CASES[tag]
    pred, gt = E.synth_eval_case(tag)
    want = g[tag + "_measures"]
    P = torch.tensor(np.stack([pred, pred * 1.1, pred])).to(DEV)
    G = torch.tensor(np.stack([gt, gt, np.zeros_like(gt)])).to(DEV)          # image 2: no valid ground truth at all
    acc = torch.tensor([1.0] * 9 + [2.0], device=DEV)
    m = evalops.compute_errors(P.unsqueeze(1), G.unsqueeze(1), 1e-3, md, ds, kb, garg, eig, eval_measures=acc)
    got = m[0].double().cpu().numpy()
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want) + 1e-7), (got, want)
    pf, valid = E.eval_prepare(pred * np.float32(1.1), gt, 1e-3, md, ds, kb, garg, eig)
    want1 = np.array(E.compute_errors(gt[valid], pf[valid]), dtype=np.float64)
    assert np.all(np.abs(m[1].double().cpu().numpy() - want1) <= 2e-5 * np.abs(want1) + 1e-7)
    assert m[2].abs().max().item() == 0.0
    exp = np.concatenate([1.0 + want + want1, [4.0]])                        # two images counted, the empty one skipped
    assert np.all(np.abs(acc.double().cpu().numpy() - exp) <= 3e-5 * np.abs(exp))
    acc2 = torch.zeros(10, device=DEV)
    evalops.compute_errors(P, G, 1e-3, md, ds, kb, garg, eig, has_valid_depth=torch.tensor([0, 1, 1]), eval_measures=acc2)
    assert acc2[9].item() == 1.0 and abs(acc2[1].item() - want1[1]) <= 3e-5 * want1[1]


def test_depth_to_uint16_bit_exact():
    """bts_test.py:179-185 payload: bit-exact against numpy on in-range values (kitti x256 over 352x1216, nyu x1000)."""
    import numpy as np
    from bts_amd import evalops
    from oracle import eval_oracle as E
    rng = np.random.RandomState(3)
    for ds, md, shape in (("kitti", 80.0, (2, 1, 352, 1216)), ("nyu", 10.0, (3, 1, 37, 53))):
        d = rng.uniform(1e-3, md, size=shape).astype(np.float32)
        d.flat[:4] = [0.0, 1.0 / 256, md, 0.99999994]
        out = evalops.depth_to_uint16(torch.tensor(d).to(DEV), ds).cpu().numpy()
        assert out.dtype == np.uint16 and np.array_equal(out, E.depth_to_uint16(d, ds))
    edge = torch.tensor([float("nan"), -1.0, 1e9, float("inf")], device=DEV)
    assert evalops.depth_to_uint16(edge, "kitti").cpu().tolist() == [0, 0, 65535, 65535]


def test_preprocess_train_vs_reference_golden(golden_dir):
    """bts_preprocess_train (crop + flip + gamma / brightness / colour + clip + ToTensor + Normalize in one kernel) on the
    16 golden cases as two batches, against the oracle that is bit-identical to the reference's methods.  Samples without
    augmentation must be bit-exact; with augmentation powf may differ from numpy's by an ulp: 2e-6 absolute on the
    normalised image.  Depth is always bit-exact."""
    import random

    import numpy as np
    from bts_amd import dataops
    from oracle import data_oracle as D
    g = np.load("%s/preprocess.npz" % golden_dir)
    img, dep = g["image_u8"], g["depth_raw"]
    for ds in ("kitti", "nyu"):
        params, want = [], []
        for seed in range(1, 9):
            p = D.draw_train_params(seed, 48, 80, 32, 64, ds)
            want.append((p, D.preprocess_train(img, dep, p, 32, 64, ds)))
            random.seed(seed)
            np.random.seed(seed)
            params.append(dataops.draw_train_params(48, 80, 32, 64, ds))
        I = torch.tensor(np.stack([img] * 8)).to(DEV)
        Z = torch.tensor(np.stack([dep] * 8)).to(DEV)
        out_i, out_d = dataops.preprocess_train(I, Z, params, 32, 64, ds)
        assert out_i.shape == (8, 3, 32, 64) and out_d.shape == (8, 1, 32, 64)
        for b, (p, (chw, d, _)) in enumerate(want):
            assert np.array_equal(out_d[b].cpu().numpy(), d), (ds, b)
            got = out_i[b].cpu().numpy()
            if p["augment"]:
                assert np.abs(got - chw).max() <= 2e-6, (ds, b, np.abs(got - chw).max())
            else:
                assert np.array_equal(got, chw), (ds, b)
